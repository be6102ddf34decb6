#!/bin/bash
# PMC passes over the kernels of one command, aggregated per kernel name (GPU box).
# usage: pmc_kernel.sh TAG KERNEL_SUBSTRING -- python3 script.py args...   (the program itself after --, no wrappers)
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=$1; SUB=$2; shift 3
OUT=$R/gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS GRBM_GUI_ACTIVE"
P3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
P4="FETCH_SIZE GRBM_GUI_ACTIVE"
P5="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
P6="TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TA_FLAT_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
P7="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE"
i=1
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do  # (the TA_/TCP_ groups P6, P7 abort rocprofv3 on this image)
  rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- "$@" > $OUT/p$i.out 2> $OUT/p$i.err; echo "pass $i done" >> $R/gpurun_out/pmc_$TAG.progress
  i=$((i+1))
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "$SUB" in r['Kernel_Name']:
            k=r['Kernel_Name'].replace('void nqa::','').split('(')[0][:80]
            agg[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k][r['Counter_Name']]+=1
with open("$OUT/summary.txt","w") as out:
  for k in sorted(agg):
    a=agg[k]; c=n[k]
    print("==",k,"launches",c['GRBM_GUI_ACTIVE']//5 if c['GRBM_GUI_ACTIVE'] else 0,file=out)
    avg={m:a[m]/c[m] for m in a}
    for m in sorted(avg): print(f"  {m:30s} {avg[m]:16.0f}",file=out)
    gui=avg.get('GRBM_GUI_ACTIVE',0)/8
    if gui:
      print(f"  cycles/dispatch (per XCD) {gui:.0f}",file=out)
      if 'SQ_VALU_MFMA_BUSY_CYCLES' in avg: print(f"  MfmaUtil {avg['SQ_VALU_MFMA_BUSY_CYCLES']/(gui*1024):.3f}",file=out)
      if 'SQ_ACTIVE_INST_VALU' in avg: print(f"  VALU active quad-cycles per SIMD-cycle {avg['SQ_ACTIVE_INST_VALU']*4/(gui*1024):.3f}",file=out)
      if 'SQ_WAVE_CYCLES' in avg: print(f"  avg waves per SIMD {avg['SQ_WAVE_CYCLES']*4/(gui*1024):.2f}; wait_any frac {avg['SQ_WAIT_ANY']/avg['SQ_WAVE_CYCLES']:.3f}; wait_inst frac {avg['SQ_WAIT_INST_ANY']/avg['SQ_WAVE_CYCLES']:.3f}; active frac {avg['SQ_ACTIVE_INST_ANY']/avg['SQ_WAVE_CYCLES']:.3f}",file=out)
      if 'TA_TA_BUSY_sum' in avg: print(f"  TA busy frac (sum over 256 TAs / (cycles*256)) {avg['TA_TA_BUSY_sum']/(gui*256):.3f}",file=out)
      if 'TCP_TCC_READ_REQ_sum' in avg and avg['TCP_TCC_READ_REQ_sum']: print(f"  TCP->TCC read latency {avg['TCP_TCC_READ_REQ_LATENCY_sum']/avg['TCP_TCC_READ_REQ_sum']:.0f} cycles per request",file=out)
      if 'SQ_LDS_IDX_ACTIVE' in avg: print(f"  LDS active frac {avg['SQ_LDS_IDX_ACTIVE']/(gui*256):.3f}",file=out)
      if 'FETCH_SIZE' in avg: print(f"  fetch (x2-corrected) {2*avg['FETCH_SIZE']*1024/1e6:.1f} MB  write {avg.get('WRITE_SIZE',0)*1024/1e6:.1f} MB  L2 hit {avg.get('TCC_HIT_sum',0)/max(avg.get('TCC_HIT_sum',0)+avg.get('TCC_MISS_sum',0),1):.3f}",file=out)
print(open("$OUT/summary.txt").read())
PY
