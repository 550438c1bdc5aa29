#!/usr/bin/env python3
"""Per-LAYER HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --dump-layers layers.json`.

usage: python tools/pmc_layers.py <fetch_counter_collection.csv> <write_counter_collection.csv> <layers.json> <out.json>

The tile configuration the autotuner assigns to a layer differs from process to process, so a per-symbol table cannot be
joined to a later bench run; the launch ORDER inside one inference call is fixed, though.  layers.json (written by bench.py)
lists the launches of one call in order; the last len(layers) `ynk::` dispatches of each trace are the final call of that
process, so row i of the tail is layer i (the kernel family is checked).  Units: the counters are KiB; on gfx950 FETCH_SIZE
under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section): hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024."""
import csv
import json
import sys


def calls_of(path, layers, reps):
    """The last `reps` calls of the trace, each aligned to the record list: a record covers one dispatch, or several
    consecutive dispatches of the same kernel family when the bracket holds more than one launch (the two sort passes)."""
    rows = [r for r in csv.DictReader(open(path)) if "ynk::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    first = family(layers[0]["kernel"])
    starts = [i for i, r in enumerate(rows) if family(r["Kernel_Name"]) == first]
    assert len(starts) >= reps, "trace shorter than %d calls" % reps
    out = []
    for k in range(reps):
        lo = starts[len(starts) - reps + k]
        hi = starts[len(starts) - reps + k + 1] if k + 1 < reps else len(rows)
        seg, j, vals = rows[lo:hi], 0, []
        known = {family(rec["kernel"]) for rec in layers}
        for i, rec in enumerate(layers):
            fam = family(rec["kernel"])
            while vals and j < len(seg) and family(seg[j]["Kernel_Name"]) not in known:      # a helper launch inside the previous bracket (nms_tile_off_kernel)
                vals[-1] += float(seg[j]["Counter_Value"]); j += 1
            assert j < len(seg) and family(seg[j]["Kernel_Name"]) == fam, (i, rec["kernel"], seg[j]["Kernel_Name"] if j < len(seg) else None)
            v = float(seg[j]["Counter_Value"]); j += 1
            nxt = family(layers[i + 1]["kernel"]) if i + 1 < len(layers) else None
            while j < len(seg) and family(seg[j]["Kernel_Name"]) == fam and nxt != fam:
                v += float(seg[j]["Counter_Value"]); j += 1
            vals.append(v)
        assert j == len(seg), "unaligned dispatches at the end of a call"
        out.append(vals)
    return out


def family(name):
    name = name.replace("void ", "").replace("ynk::", "")
    name = name.split("<")[0].split("(")[0].strip()
    return "pointwise_gemm" if name in ("gemm_conv_kernel", "gemm_direct_kernel", "gemm_split_kernel", "pw_pipe_kernel") else name   # autotuned per process


def main():
    fetch_csv, write_csv, layers_json, out_json = sys.argv[1:5]
    layers = json.load(open(layers_json))
    reps = 3
    f, w = calls_of(fetch_csv, layers, reps), calls_of(write_csv, layers, reps)
    out = {"_meta": {"workload": layers and layers[0].get("workload"), "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes",
                     "calls_averaged": reps, "source_hash": layers and layers[0].get("source_hash")}}
    print("| # | layer | kernel | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB | HBM MB = (2F+W)*1024 | algorithmic MB |\n|---|---|---|---|---|---|---|")
    for i, rec in enumerate(layers):
        fk = sum(c[i] for c in f) / reps
        wk = sum(c[i] for c in w) / reps
        hbm = (2 * fk + wk) * 1024
        out["%d:%s" % (i, rec["layer"])] = {"kernel": rec["kernel"], "fetch_kib": round(fk, 1), "write_kib": round(wk, 1),
                                           "hbm_bytes": round(hbm), "alg_bytes": round(rec["bytes"])}
        print("| %d | %s | `%s` | %.0f | %.0f | %.2f | %.2f |" % (i, rec["layer"], rec["kernel"], fk, wk, hbm / 1e6, rec["bytes"] / 1e6))
    json.dump(out, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
