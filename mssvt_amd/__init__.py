"""mssvt_amd -- MI355X-native MsSVT backbone (see README.md / DESIGN.md)."""
import os as _os

# Kernel arguments in device memory instead of host-coherent memory (a ROCm runtime setting that is read when HIP
# initialises, i.e. it must be in the environment before the first GPU call of the process): a frame is ~34 short launches
# with argument blocks of up to 1 KB (four head groups' pointers in one struct), and every launch otherwise starts with
# the command processor fetching them across the host link.  Measured on one box (bench.py, configs[1]): 0.80 -> 0.76 ms
# per frame.  An explicit setting in the environment wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
