// pya_version(): the library names the source tree it was linked from.
//
// build.py compiles this file at every link with PYA_TREE_DIGEST = SHA-256 over every file under
// pyascore_amd/csrc/ and include/ (build.py:tree_digest).  tests/test_c_abi.py and the -m gpu suite
// recompute the digest from the tree they see and compare, so a stale object cannot pass for HEAD.
#include "../../include/pyascore_hip.h"

#ifndef PYA_TREE_DIGEST
#error "version.cpp is compiled by pyascore_amd/build.py, which passes -DPYA_TREE_DIGEST"
#endif

extern "C" const char *pya_version(void) { return "pyascore_hip 0.4.0 (gfx950) src=" PYA_TREE_DIGEST; }
