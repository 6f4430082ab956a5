"""Turns the two per-kernel PMC summaries written by tools/profile_round.sh into profiles/<tag>_traffic.json.

FETCH_SIZE / WRITE_SIZE are reported in KiB-sized units (x1024 -> bytes).  On gfx950 FETCH_SIZE counts half of a
wide coalesced read stream (MI355X_MICROARCH.md, HBM / rocprofv3 section) and is doubled; WRITE_SIZE is taken as
reported.  The PMC command is `bench.py --steps 1 --warmup 0 --plms-steps 10 --no-vae` = 13 executions of the
2B UNet forward (alpha_type [0.3, 0, 0.7] over 10 steps: 3 fuser-on + PLMS second-order extra call, ...), so the
per-launch figure divides by the number of forward launches passed on the command line (default 13).
usage: python tools/traffic_from_pmc.py <fetch.csv> <write.csv> <out.json> [forwards] [git_head]
(git_head: the tree the PMC passes ran on; bench.py prints it as roofline.traffic_age so a stale file is visible)
"""
import csv
import json
import sys


def total(path, col):
    s = 0.0
    for r in csv.DictReader(open(path)):
        k = r["kernel"]
        # this repo's HIP kernels only: they live in anonymous namespaces of the library's translation units.  torch's own kernels do too
        # (at::native::(anonymous namespace)::distribution_elementwise... = the weight-init RNG of bench.py: +0.39 GB "per forward" in the
        # round-4 file), so everything under at:: / c10:: / rocprim / hipcub / thrust is excluded by name
        if any(t in k for t in ("at::", "c10::", "rocprim", "hipcub", "thrust::", "void at_")):
            continue
        if "anonymous namespace" in k or "_GLOBAL__N_" in k:
            s += float(r[col])
    return s * 1024.0


def main():
    fetch, write, out = sys.argv[1:4]
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 13
    head = sys.argv[5] if len(sys.argv) > 5 else None
    f = total(fetch, "sum_FETCH_SIZE") / n
    w = total(write, "sum_WRITE_SIZE") / n
    doc = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_round.sh) over "
                  "`bench.py --steps 1 --warmup 0 --plms-steps 10 --no-vae` = %d executions of the 2B UNet forward; "
                  "HIP kernels of this repo only" % n,
        "fetch_bytes_per_forward_reported": f,
        "write_bytes_per_forward_reported": w,
        "gfx950_fetch_correction": 2.0,
        "traffic_bytes_per_forward": 2.0 * f + w,
        "git_head": head,
        "note": "FETCH_SIZE on gfx950 reports half of a wide coalesced read stream (MI355X_MICROARCH.md HBM section) "
                "-> doubled; WRITE_SIZE uncalibrated, taken as reported; Infinity-Cache hits are included in both",
    }
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
