#!/bin/bash
# PMC passes on Dynamics.forward (uniform n = 50, B = 256: the pair loop's steady state), forward kernel only:
#   scripts/r4/pmc_forward.sh <precision> -> gpurun_out/r4/pmc_forward_<precision>.txt
prec=${1:-f16x3}
export TMPDIR=/tmp
ROOT=$(pwd)
out=$ROOT/gpurun_out/r4/pmc_forward_$prec.txt
: > $out
run() {
  name=$1; shift
  ( cd /tmp && rm -rf /tmp/rp_$name && timeout 300 rocprofv3 --pmc "$@" -d /tmp/rp_$name --output-format csv -- python $ROOT/scripts/time_forward.py --precision $prec ) > /tmp/rp_$name.log 2>&1
  f=$(find /tmp/rp_$name -name '*counter_collection.csv' | head -n 1)
  python3 - "$f" >> $out <<'PY'
import csv, sys
s, c = {}, {}
for r in csv.DictReader(open(sys.argv[1])):
    if 'egnn_forward_fc_kernel' not in r['Kernel_Name']:
        continue
    s[r['Counter_Name']] = s.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    c[r['Counter_Name']] = c.get(r['Counter_Name'], 0) + 1
for k in sorted(s):
    print(f'{k:32s} {s[k] / c[k]:18.1f}   (average over {c[k]} launches)')
PY
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
run b SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32
run c SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
cat $out
