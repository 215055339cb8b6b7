"""Developer tool: measured ceilings of the box (fp32 MFMA rate, HBM stream bandwidth) next to the guide's peaks.

    python tools/peaks/run.py            # builds tools/peaks/libpeaks.so with hipcc if missing, prints JSON
"""
import ctypes
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libpeaks.so")


def build():
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "peaks.hip")):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
                               os.path.join(HERE, "peaks.hip"), "-o", SO])


def main():
    build()
    lib = ctypes.CDLL(SO)
    lib.peaks_mfma_f32.restype = ctypes.c_double
    lib.peaks_mfma_f32.argtypes = [ctypes.c_int] * 5
    lib.peaks_hbm.restype = ctypes.c_double
    lib.peaks_hbm.argtypes = [ctypes.c_longlong, ctypes.c_int, ctypes.c_int]
    res = {"mfma_f32_tflops": {}, "hbm_gbs": {}}
    for wps in (1, 2, 4):
        for chains in (4, 8):
            res["mfma_f32_tflops"]["%d waves/SIMD, %d chains" % (wps, chains)] = round(lib.peaks_mfma_f32(256, wps, chains, 20000, 0), 1)
            res["mfma_f32_tflops"]["%d waves/SIMD, %d chains, LDS weight reads" % (wps, chains)] = round(lib.peaks_mfma_f32(256, wps, chains, 1000, 1), 1)
    for mb in (256, 2048):
        for mode, name in ((0, "copy"), (1, "read"), (2, "write")):
            for grid in (2048, 8192):
                res["hbm_gbs"]["%s %d MiB, %d blocks" % (name, mb, grid)] = round(lib.peaks_hbm(mb << 20, mode, grid), 0)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
