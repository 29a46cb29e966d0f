"""Condense the MFMA counter pass of `bench.py --workload corr` (rocprofv3 --pmc, its own run) and the FETCH/WRITE passes into
profiles/TAG_corr_mfma.txt (per-kernel table) + TAG_corr_mfma.json (what bench.py's extra.corr.roofline reads).
  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs
(MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, 16 per v_mfma_f32_16x16x32_bf16; GRBM_GUI_ACTIVE is the sum over the XCDs)."""
import csv, json, re, sys, collections


def per_kernel(path):
    disp = collections.defaultdict(float)
    with open(path) as f:
        for r in csv.DictReader(f):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            disp[(name, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for (name, _, c), v in disp.items(): acc[name][c].append(v)
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def main():
    mf, fetch, write, out, tag = per_kernel(sys.argv[1]), per_kernel(sys.argv[2]), per_kernel(sys.argv[3]), sys.argv[4], sys.argv[5]
    res = {"_note": "mean per launch over `bench.py --workload corr` under rocprofv3 --pmc (separate passes for the SQ/GRBM counters, FETCH_SIZE and "
                    "WRITE_SIZE); mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); traffic = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB "
                    "(gfx950 FETCH correction, MI355X_MICROARCH.md HBM section)", "_source": tag, "kernels": {}}
    lines = []
    for k, c in sorted(mf.items()):
        if not k.startswith("k_corr"): continue
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        util = busy / (1024.0 * cyc) if cyc else None
        fk = fetch.get(k, {}).get("FETCH_SIZE", 0.0); wk = write.get(k, {}).get("WRITE_SIZE", 0.0)
        res["kernels"][k] = {**{n: v for n, v in c.items()}, "kernel_cycles": cyc, "mfma_util": util,
                             "fetch_size_kib": fk, "write_size_kib": wk, "traffic_bytes": int(2 * fk * 1024 + wk * 1024)}
        lines.append(f"{k:24s} " + " ".join(f"{n}={v:,.0f}" for n, v in sorted(c.items())) + f" kernel_cycles={cyc:,.0f} mfma_util={util:.4f} "
                     f"FETCH_SIZE_KiB={fk:,.0f} WRITE_SIZE_KiB={wk:,.0f}")
    main_k = max(res["kernels"], key=lambda k: res["kernels"][k]["kernel_cycles"]) if res["kernels"] else None
    if main_k:
        res["kernel"] = main_k
        res["mfma_util"] = res["kernels"][main_k]["mfma_util"]
        for n in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_MFMA", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
            if n in res["kernels"][main_k]: res[n] = res["kernels"][main_k][n]
    open(f"{out}/{tag}_corr_mfma.txt", "w").write("\n".join(lines) + "\n")
    json.dump(res, open(f"{out}/{tag}_corr_mfma.json", "w"), indent=1)


if __name__ == "__main__":
    main()
