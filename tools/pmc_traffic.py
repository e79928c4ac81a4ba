#!/usr/bin/env python
"""Merge the three PMC pass summaries of tools/pmc.sh (fetch, write, mfma) into profiles/<round>/pmc_traffic_<tag>.json:
tools/pmc_traffic.py <fetch.summary> <write.summary> <mfma.summary> <out.json>
hbm_bytes_per_launch = 2 * FETCH_SIZE KiB * 1024 + WRITE_SIZE KiB * 1024 (gfx950 read correction, MI355X_MICROARCH.md);
mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)."""
import json, sys
fetch, write, mfma = (json.load(open(p)) for p in sys.argv[1:4])
out = {'command': 'rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 '
                  'GRBM_GUI_ACTIVE> --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --new-tokens 12 --no-cpu-baseline '
                  '(three separate passes, tools/pmc.sh)',
       'units': 'FETCH_SIZE / WRITE_SIZE in KiB per launch (average over launches); hbm_bytes = 2*FETCH*1024 + WRITE*1024 (gfx950: '
                'FETCH_SIZE tallies 128-B read requests at 64 B, MI355X_MICROARCH.md HBM section); mfma_util = '
                'SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)',
       'kernels': {}}
for k, v in fetch.items():
    if 'FETCH_SIZE' not in v:
        continue
    f = v['FETCH_SIZE']['avg']
    w = write.get(k, {}).get('WRITE_SIZE', {}).get('avg', 0.0)
    rec = {'launches': v['FETCH_SIZE']['launches'], 'fetch_kib_avg': f, 'write_kib_avg': w, 'hbm_bytes_per_launch': 2 * f * 1024 + w * 1024}
    m = mfma.get(k, {})
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in m and 'GRBM_GUI_ACTIVE' in m and m['GRBM_GUI_ACTIVE']['avg'] > 0:
        rec['mfma_util'] = m['SQ_VALU_MFMA_BUSY_CYCLES']['avg'] / (m['GRBM_GUI_ACTIVE']['avg'] / 8 * 1024)
        if 'SQ_INSTS_VALU_MFMA_MOPS_BF16' in m:
            rec['mfma_mops_bf16_avg'] = m['SQ_INSTS_VALU_MFMA_MOPS_BF16']['avg']
    out['kernels'][k] = rec
json.dump(out, open(sys.argv[4], 'w'), indent=1)
print('wrote', sys.argv[4], len(out['kernels']), 'kernels')
