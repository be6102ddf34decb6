#!/bin/bash
# PMC passes over one conv layer (GPU box).  usage: pmc_conv.sh LAYER VARIANT TAG [256|1080] [REPS]
R=${GRAFT_REPO_ROOT:-$PWD}; L=$1; V=$2; TAG=$3; WHICH=${4:-256}; REPS=${5:-8}
OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS GRBM_GUI_ACTIVE"
P3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"
i=1
P4="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum GRBM_GUI_ACTIVE"
for P in "$P1" "$P2" "$P3" "$P4"; do
  rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 $R/tools/gpu_one_layer.py $L $V $WHICH $REPS > /dev/null 2> $OUT/p$i.err
  i=$((i+1))
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(float); n=collections.Counter()
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'igemm' in r['Kernel_Name'] or 'regw' in r['Kernel_Name']:
            agg[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for k in sorted(agg): print(f"{k:32s} {agg[k]/n[k]:16.0f}  (n={n[k]})")
gui=agg['GRBM_GUI_ACTIVE']/n['GRBM_GUI_ACTIVE']/8
print("cycles/dispatch (per XCD)", gui)
if 'SQ_VALU_MFMA_BUSY_CYCLES' in agg: print("MfmaUtil %.3f" % (agg['SQ_VALU_MFMA_BUSY_CYCLES']/n['SQ_VALU_MFMA_BUSY_CYCLES']/(gui*1024)))
if 'TCC_HIT_sum' in agg: print("L2 hit rate %.3f" % (agg['TCC_HIT_sum']/max(agg['TCC_HIT_sum']+agg['TCC_MISS_sum'],1)))
if 'SQ_LDS_IDX_ACTIVE' in agg: print("LDS active frac %.3f conflict frac %.3f" % (agg['SQ_LDS_IDX_ACTIVE']/n['SQ_LDS_IDX_ACTIVE']/(gui*256), agg['SQ_LDS_BANK_CONFLICT']/max(agg['SQ_LDS_IDX_ACTIVE'],1)))
PY
