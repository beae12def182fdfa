"""``python -m pyascore ...``: the reference's command line, served by :mod:`pyascore_amd.__main__`."""
from pyascore_amd.__main__ import main

if __name__ == "__main__":
    main()
