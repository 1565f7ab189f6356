"""profiles/stage_a_pmc.json from the summary scripts/pmc_stage_a.sh prints (development aid).  usage: stage_a_pmc_json.py SUMMARY.txt OUT.json [cells] [order]"""
import json, re, sys
txt = open(sys.argv[1]).read()
cells = int(sys.argv[3]) if len(sys.argv) > 3 else 48
order = int(sys.argv[4]) if len(sys.argv) > 4 else 5
m = re.search(r"exa::(dg_stage_a_\w+<[^\n]*>)\n((?:   .*\n)+)", txt)
name, body = m.group(1), m.group(2)
c = {k: float(v) for k, v in re.findall(r"(\w+)\s+n=\d+ mean=([0-9.e+]+)", body)}
waves = c["SQ_WAVES"]
simd_time = c["SQ_WAVE_CYCLES"] * 4 / (waves / 1024)          # cycles a SIMD had this kernel's waves resident, summed over the 1024 SIMDs
cu_time = c["SQ_WAVE_CYCLES"] * 4 / (waves / 256)
rec = {"kernel": name, "cells": cells, "order": order, "counters": c,
       "valu_issue_frac": c["SQ_INSTS_VALU"] * 4 / simd_time, "lds_array_busy_frac": c["SQ_LDS_IDX_ACTIVE"] / cu_time,
       "lds_bank_conflict_frac": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], "wait_any_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
       "wait_inst_any_frac": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], "mfma_insts": 0,
       "how": "scripts/pmc_stage_a.sh (rocprofv3 --pmc, four separate passes, no trace domains) + scripts/stage_a_pmc_json.py; valu_issue_frac = SQ_INSTS_VALU x 4 "
              "cycles over the time a SIMD had the kernel's waves resident (SQ_WAVE_CYCLES per wave, quad-cycles); lds_array_busy_frac = SQ_LDS_IDX_ACTIVE per CU over the same time"}
json.dump(rec, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: rec[k] for k in ("kernel", "valu_issue_frac", "lds_array_busy_frac", "lds_bank_conflict_frac", "wait_any_frac", "wait_inst_any_frac")}))
