#!/usr/bin/env python3
"""profiles/<round>/pmc_select.txt from the per-regime counter summaries tools/exp/pmc_all.sh left in gpurun_out/pmc_<tag>_<suffix>.txt
    python tools/exp/pmc_select_report.py r05 r5"""
import glob, os, re, sys

rnd, suffix = sys.argv[1], sys.argv[2]
out = open(f"profiles/{rnd}/pmc_select.txt", "w")
out.write(f"per-cell select, SQ / HBM counters per regime (rocprofv3 --pmc, one counter group per pass: tools/exp/pmc_tile.sh via tools/exp/pmc_all.sh), MI355X, round {rnd[1:].lstrip('0')}\n"
          "|N(0,1)| scores [n, M], M = 2^k + 64 cells (no power-of-two row pitch), 10 ranks, one launch per pass; counters are sums over all SEs / XCDs of one launch.\n"
          "derived: VALU per element = SQ_INSTS_VALU x 64 lanes / (n x M);  VALU busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x SQ_BUSY_CYCLES / 32);\n"
          "         LDS busy = SQ_LDS_IDX_ACTIVE / (256 CUs x SQ_BUSY_CYCLES / 32);  FETCH = FETCH_SIZE KiB x 1024 x 2 (gfx950 correction) / (4 n M);\n"
          "         waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES.\n"
          "(timings under the counter passes are 10-20 % slower than free-running: select_scan.txt has those.)\n\n")
files = sorted(glob.glob(f"gpurun_out/pmc_n*_{suffix}.txt"), key=lambda f: int(re.search(r"pmc_n(\d+)_", f).group(1)))
for f in files:
    txt = open(f).read().splitlines()
    kern, vals, nm = None, {}, None
    for l in txt:
        if l.startswith("void") or "kth_" in l and "mean=" not in l and not l.startswith("n="):
            kern = re.sub(r"void |\(anonymous namespace\)::", "", l).strip()
        m = re.match(r"\s+(\w+)\s+n=\d+ mean=([\d.e+]+)", l)
        if m:
            vals[m.group(1)] = float(m.group(2))
        m = re.match(r"n=(\d+) M=(\d+)", l)
        if m:
            nm = (int(m.group(1)), int(m.group(2)))
    if not (kern and vals and nm):
        continue
    n, M = nm
    out.write(f"n = {n}, M = {M}  ({4 * n * M / 1e9:.2f} GB)   {kern}\n")
    for k, v in sorted(vals.items()):
        out.write(f"   {k:26s} {v:.4g}\n")
    el = n * M
    busy = vals.get("SQ_BUSY_CYCLES", 0) / 32
    d = []
    if "SQ_INSTS_VALU" in vals: d.append(f"VALU per element {vals['SQ_INSTS_VALU'] * 64 / el:.1f}")
    if "SQ_INSTS_SALU" in vals: d.append(f"SALU per element {vals['SQ_INSTS_SALU'] * 64 / el:.1f}")
    if "SQ_INSTS_LDS" in vals: d.append(f"LDS instructions per element {vals['SQ_INSTS_LDS'] * 64 / el:.2f}")
    if busy and "SQ_ACTIVE_INST_VALU" in vals: d.append(f"VALU busy {vals['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * busy):.2f}")
    if busy and "SQ_LDS_IDX_ACTIVE" in vals: d.append(f"LDS busy {vals['SQ_LDS_IDX_ACTIVE'] / (256 * busy):.2f}")
    if vals.get("SQ_LDS_IDX_ACTIVE"): d.append(f"LDS bank-conflict share {vals.get('SQ_LDS_BANK_CONFLICT', 0) / vals['SQ_LDS_IDX_ACTIVE']:.2f}")
    if vals.get("SQ_WAVE_CYCLES"): d.append(f"waiting {vals.get('SQ_WAIT_ANY', 0) / vals['SQ_WAVE_CYCLES']:.2f}")
    if "FETCH_SIZE" in vals: d.append(f"FETCH {vals['FETCH_SIZE'] * 1024 * 2 / (4 * el):.2f} x the scores")
    out.write("   -> " + "   ".join(d) + "\n\n")
out.close()
print(open(f"profiles/{rnd}/pmc_select.txt").read())
