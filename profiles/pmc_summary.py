"""Summarise rocprofv3 --pmc passes (csv output): per kernel name and counter, launches / mean / sum.

    python profiles/pmc_summary.py <dir> [<dir> ...] [--match SUBSTR] [--traffic-json OUT]

--traffic-json writes {kernel: {"hbm_bytes_per_launch", "fetch_bytes", "write_bytes", "launches"}} from FETCH_SIZE /
WRITE_SIZE passes: rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes for wide
(16 B / lane) streaming reads, so it is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-byte stores.
bench.py reads that file for `roofline.traffic`."""
import collections
import csv
import glob
import json
import os
import sys


def norm_name(k):
    """rocprofv3 leaves names with a __bf16 template argument mangled: spell every kernel the way the library's
    coma_last_kernel() does ("conv_mfma_halo2_k<2, 32, 1, 1, __bf16>")."""
    import re
    k = k.strip()
    m = re.match(r"_Z\d+([A-Za-z0-9_]+?)I((?:Li\d+E|DF16b|f|Lb[01]E)+)E", k)
    if m:
        args = []
        for t in re.findall(r"Li(\d+)E|(DF16b)|(f)|Lb([01])E", m.group(2)):
            args.append(t[0] or ("__bf16" if t[1] else "float" if t[2] else t[3]))
        return f"{m.group(1)}<{', '.join(args)}>"
    k = k.replace("void ", "")
    return k.split("(")[0]


def main():
    args = sys.argv[1:]
    match, tj, dirs = None, None, []
    while args:
        a = args.pop(0)
        if a == "--match":
            match = args.pop(0)
        elif a == "--traffic-json":
            tj = args.pop(0)
        else:
            dirs.append(a)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = row.get("Kernel_Name", "?")
                    if match and match not in k:
                        continue
                    acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc, key=lambda k: -sum(len(v) for v in acc[k].values())):
        print(k[:150])
        for c in sorted(acc[k]):
            v = acc[k][c]
            print(f"   {c:28s} n={len(v):4d} avg={sum(v) / len(v):.4g} sum={sum(v):.4g}")
    if tj:
        out = {}
        for k, cs in acc.items():
            if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
                f = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]) * 1024 * 2.0
                w = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]) * 1024
                name = norm_name(k)
                out[name] = {"hbm_bytes_per_launch": f + w, "fetch_bytes": f, "write_bytes": w,
                             "launches": len(cs["FETCH_SIZE"]), "note": "FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read correction) + WRITE_SIZE KiB x 1024"}
        json.dump(out, open(tj, "w"), indent=1, sort_keys=True)
        print("wrote", tj, len(out), "kernels")


if __name__ == "__main__":
    main()
