"""mssvt_amd -- MI355X-native MsSVT backbone (see README.md / DESIGN.md)."""
import os as _os


def use_device_kernargs():
    """Ask the ROCm runtime to keep kernel arguments in device memory instead of host-coherent memory
    (HIP_FORCE_DEV_KERNARG=1): a frame is ~27 short launches with argument blocks of up to 1 KB (four head groups'
    pointers in one struct), and every launch otherwise starts with the command processor fetching them across the host
    link -- 0.80 -> 0.76 ms per frame on one box (bench.py, configs[1]).  The runtime reads the variable when HIP
    initialises, so this must run before the first GPU call of the PROCESS; it changes the environment of the host
    application and of its children, which is why importing the package does not do it: bench.py and the tools call it,
    an application opts in here or with MSSVT_DEV_KERNARG=1.  An explicit HIP_FORCE_DEV_KERNARG in the environment wins."""
    _os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")


if _os.environ.get("MSSVT_DEV_KERNARG", "0") == "1":
    use_device_kernargs()
