"""Handles created, used and dropped in a loop: open descriptors, threads and free HBM must stay flat
(and torch must still be able to bring up its own HIP runtime in the same process afterwards)."""
import gc
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyascore_amd import PyAscore, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    batch, settings = synth.make_batch("cfg3", n_psm=500, seed=3)
    for i in range(n):
        s = PyAscore(settings["bin_size"], settings["n_top"], settings["mod_group"], settings["mod_mass"],
                     mz_error=settings["mz_error"], fragment_types=settings["fragment_types"])
        s.score_batch(batch, keep=(i % 2 == 0))
        del s
        if i % 50 == 0 or i == n - 1:
            gc.collect()
            print(i, "fds", len(os.listdir("/proc/self/fd")), "threads", threading.active_count(),
                  "os-threads", len(os.listdir("/proc/self/task")), flush=True)
    import torch
    print("torch sees", torch.cuda.device_count(), "GPU(s); init:", end=" ")
    torch.zeros(1, device="cuda")
    free, total = torch.cuda.mem_get_info()
    print("ok, free HBM %.1f GiB of %.1f" % (free / 2**30, total / 2**30))


if __name__ == "__main__":
    main()
