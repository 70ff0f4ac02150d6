#!/usr/bin/env python3
"""What a gfx950 SIMD issues per cycle, by instruction kind and resident waves per SIMD
(raytracer_amd/libvalu_peak.so, built from tools/valu_peak.hip).  Needs a GPU.

    python tools/valu_peak.py [--iters 4000] [--json out.json]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "raytracer_amd", "libvalu_peak.so")


def load():
    lib = C.CDLL(SO)
    lib.valu_peak_kind_name.restype = C.c_char_p
    lib.valu_peak_run.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    return lib


def measure(lib, kind, waves, iters=4000):
    out = (C.c_double * 4)()
    rc = lib.valu_peak_run(kind, waves, iters, out)
    if rc != 0:
        raise RuntimeError(f"valu_peak_run failed with HIP error {rc}")
    return dict(instr_per_cycle_per_simd=out[0], instr_per_cycle_per_simd_median=out[1], clock_mhz=out[2], chip_instr_per_s=out[3])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=4000)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    lib = load()
    rows = {}
    print(f"{'kind':28s}" + "".join(f"{w} waves/SIMD".rjust(16) for w in (1, 2, 4, 8)) + "   (VALU instr / cycle / SIMD; cycles per instr in brackets)")
    for kind in range(lib.valu_peak_kinds()):
        name = lib.valu_peak_kind_name(kind).decode()
        rows[name] = {}
        line = f"{name:28s}"
        for waves in (1, 2, 4, 8):
            r = measure(lib, kind, waves, args.iters)
            rows[name][str(waves)] = r
            ipc = r["instr_per_cycle_per_simd"]
            line += f"{ipc:9.3f} [{1 / ipc:4.2f}]"
        line += f"   clock {r['clock_mhz']:.0f} MHz, chip {r['chip_instr_per_s'] / 1e9:.0f} Ginstr/s at 8"
        print(line, flush=True)
    if args.json:
        with open(args.json, "w") as fh:
            json.dump(rows, fh, indent=1)


if __name__ == "__main__":
    main()
