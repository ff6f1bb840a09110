"""Build profiles/rNN_conv_hbm_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of

    bench.py --steps 1 --warmup 0 --sampling-steps 2 --no-cpu-baseline --no-roofline

Usage: python tools/hbm_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
Corrections as MI355X_MICROARCH.md's HBM/rocprofv3 section prescribes: separate passes, values in KB, FETCH_SIZE x2 on gfx950.
"""
import csv
import json
import sys


def avg(path, counter):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if r["Counter_Name"] != counter or not ("conv3x3_ws_kernel" in k or "conv3x3_c64_kernel" in k or "conv3x3_w256_kernel" in k or "conv_igemm_kernel" in k or "conv3x3_halo_kernel" in k):
            continue
        tot += float(r["Counter_Value"])
        n += 1
    return tot / max(1, n), n


f, nf = avg(sys.argv[1], "FETCH_SIZE")
w, nw = avg(sys.argv[2], "WRITE_SIZE")
out = {
    "kernel_class": "conv3x3_ws_kernel + conv3x3_c64_kernel + conv3x3_w256_kernel + conv_igemm_kernel (all MFMA convolution launches)",
    "launches": nf,
    "fetch_size_kb_avg": f,
    "write_size_kb_avg": w,
    "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
    "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B for wide coalesced reads, MI355X_MICROARCH.md HBM section); "
                  "WRITE_SIZE as reported; KB -> bytes x1024",
    "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace ... ; rocprofv3 --pmc WRITE_SIZE --kernel-trace ... -- python bench.py "
               "--steps 1 --warmup 0 --streams 1 --sampling-steps 2 --no-cpu-baseline --no-roofline --no-configs4 (separate passes)",
    "round": "r05",
    "kernel_sources_sha256": __import__("hashlib").sha256(b"".join(
        open(__import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..",
                                        "pointreggpt_amd", "csrc", f), "rb").read()
        for f in ("conv_ws.hip", "conv_c64.hip", "conv_w256.hip", "conv.hip", "conv.h", "common.h"))).hexdigest(),
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
