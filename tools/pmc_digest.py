#!/usr/bin/env python3
"""Digest of tools/pmc_summary.py output for the traversal kernels: per k_list / k_pc kernel the counters behind the
VALU-issue analysis of DESIGN.md section 4. usage: pmc_digest.py <pmc_summary.txt>"""
import re, sys
cur, data = None, {}
for line in open(sys.argv[1]):
    if not line.startswith("   "):
        cur = line.strip()
        data[cur] = {}
    else:
        m = re.match(r"\s+(\S+)\s+n=(\d+)\s+avg=(\S+)", line)
        if m:
            data[cur][m.group(1)] = float(m.group(3))
tot = {}
print("%-52s %10s %9s %9s %8s %8s %8s %7s %7s %9s" % ("kernel", "VALU inst", "trans", "LDS inst", "active%", "istall%", "waitcnt%", "ldsbc%", "L2hit%", "GRBM cyc"))
for k, d in data.items():
    if not any(t in k for t in ("k_list<", "k_pc<", "k_super", "k_common<", "k_lists<", "k_dense<", "k_combine<")):
        continue
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    row = (k[:52].replace("void rk::", "").replace("(rk::kparams<float>", ""), d.get("SQ_INSTS_VALU", 0), d.get("SQ_INSTS_VALU_TRANS", 0), d.get("SQ_INSTS_LDS", 0),
           100 * d.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * d.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * d.get("SQ_WAIT_ANY", 0) / wc,
           100 * d.get("SQ_LDS_BANK_CONFLICT", 0) / max(d.get("SQ_LDS_IDX_ACTIVE", 0), 1),
           100 * d.get("TCC_HIT_sum", 0) / max(d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0), 1), d.get("GRBM_GUI_ACTIVE", 0) / 8)
    print("%-52s %10.4g %9.3g %9.3g %8.1f %8.1f %8.1f %7.1f %7.1f %9.4g" % row)
    for i, name in enumerate(("valu", "trans", "lds")):
        tot[name] = tot.get(name, 0) + row[1 + i]
    tot["grbm"] = tot.get("grbm", 0) + row[-1]
print("sum over the traversal kernels of one step: VALU wave-instructions %.4g (of which transcendental %.3g), LDS %.3g, "
      "GRBM_GUI_ACTIVE / 8 = %.4g cycles (kernels serialised by the profiler)" % (tot.get("valu", 0), tot.get("trans", 0), tot.get("lds", 0), tot.get("grbm", 0)))
print("columns: active / istall / waitcnt = SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY / SQ_WAIT_ANY over SQ_WAVE_CYCLES (share of a resident "
      "wave's lifetime spent issuing / stalled on issue (pipe busy with other waves, dependencies) / parked on s_waitcnt); "
      "ldsbc = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE")
