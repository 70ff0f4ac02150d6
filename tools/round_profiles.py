#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/profile_counters.sh (+ profile_extra.sh) into the
files kept under profiles/:  <tag>_kernel_stats.csv (the kernel trace's per-kernel table),
<tag>_render_counters.json (per-launch counter means of the render kernel) and
<tag>_hbm_traffic.json (HBM bytes per launch: FETCH_SIZE doubled, because gfx950 tallies a
128-byte read request as 64 bytes, plus WRITE_SIZE; both counters are in KB).
The tag is r<round> for the headline workload and r<round>_<workload> for another (bench.py looks them up so).
Usage: tools/round_profiles.py <rocprof outdir> <tag> [kernel name substring]"""
import csv, glob, json, os, shutil, subprocess, sys

d, tag = sys.argv[1], sys.argv[2]
kernel = sys.argv[3] if len(sys.argv) > 3 else "renderKernel"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
summary = json.loads(subprocess.check_output([sys.executable, os.path.join(root, "tools", "summarize_counters.py"), d, kernel]))
# the library the counters were taken on (the tree that went to the GPU box is this one): bench.py reports
# "counters_stale" when the library it runs is another
import hashlib
with open(os.path.join(root, "raytracer_amd", "librtx_hip.so"), "rb") as fh:
    summary["library_sha256"] = hashlib.sha256(fh.read()).hexdigest()
with open(os.path.join(prof, f"{tag}_render_counters.json"), "w") as fh:
    json.dump(summary, fh, indent=1)
ks = glob.glob(os.path.join(d, "trace", "*kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], os.path.join(prof, f"{tag}_kernel_stats.csv"))
c = summary["counters_per_launch_mean"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    name = summary.get("kernel_stats", {}).get("Name", kernel)
    with open(os.path.join(prof, f"{tag}_hbm_traffic.json"), "w") as fh:
        json.dump({"source": f"profiles/{tag}_render_counters.json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, bench.py --steps 5 --warmup 1)",
                   "kernel": name, "fetch_size_kb_per_launch": c["FETCH_SIZE"], "write_size_kb_per_launch": c["WRITE_SIZE"],
                   "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B); WRITE_SIZE as reported; both are KB",
                   "bytes_per_launch": int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)}, fh, indent=1)
print(json.dumps({k: summary.get(k) for k in ("kernel_stats",)}, indent=1))
