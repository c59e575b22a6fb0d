#!/bin/bash
# SQ / LDS counters per kernel over a few bench.py steps (two rocprofv3 --pmc passes, counters only):
#   tools/pmc_step.sh <outdir> [bench.py args...]
set -e
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --windows 1 --no-cpu-baseline --no-fp32-compare --no-kernel-timing --no-other-configs $*"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES"
P2="SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
timeout -k 10 300 rocprofv3 --pmc $P1 --output-format csv -d $OUT/p1 -o p -- python3 $R/bench.py $ARGS > $OUT/p1.log 2>&1
echo "pass 1 done"
timeout -k 10 300 rocprofv3 --pmc $P2 --output-format csv -d $OUT/p2 -o p -- python3 $R/bench.py $ARGS > $OUT/p2.log 2>&1
echo "pass 2 done"
python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("$OUT/p1","$OUT/p2"):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].replace("void ","").split("(")[0][:48]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows=[]
for k,c in acc.items():
    m={n: sum(v)/len(v) for n,v in c.items()}
    if "SQ_WAVE_CYCLES" not in m or m["SQ_WAVE_CYCLES"] < 1e6: continue
    wc=m["SQ_WAVE_CYCLES"]
    rows.append((wc*len(c["SQ_WAVE_CYCLES"]), k, m))
rows.sort(reverse=True)
with open("$OUT/summary.txt","w") as f:
    f.write("kernel; launches; waves; per wave-cycle: active_any wait_any wait_inst_any valu_active lds_active wait_lds mfma_busy/4; insts per wave: valu lds mfma; lds bank conflict share\n")
    for tot,k,m in rows[:28]:
        wc=m["SQ_WAVE_CYCLES"]; w=max(m.get("SQ_WAVES",1),1)
        g=lambda n: m.get(n,0.0)
        line=(f"{k:48s} n={len(acc[k]['SQ_WAVE_CYCLES']):4d} waves={w:6.0f}  any {g('SQ_ACTIVE_INST_ANY')/wc:5.2f} wait {g('SQ_WAIT_ANY')/wc:5.2f} waitI {g('SQ_WAIT_INST_ANY')/wc:5.2f} "
              f"valu {g('SQ_ACTIVE_INST_VALU')/wc:5.2f} lds {g('SQ_ACTIVE_INST_LDS')/wc:5.2f} waitlds {g('SQ_WAIT_INST_LDS')/wc:5.2f} mfma {g('SQ_VALU_MFMA_BUSY_CYCLES')/4/wc:5.2f} | "
              f"VALU {g('SQ_INSTS_VALU')/w:7.0f} LDS {g('SQ_INSTS_LDS')/w:6.0f} MFMA {g('SQ_INSTS_MFMA')/w:6.0f} | conflict {g('SQ_LDS_BANK_CONFLICT')/max(g('SQ_LDS_IDX_ACTIVE'),1):4.2f}")
        f.write(line+"\n"); print(line)
PY
rm -rf $OUT/p1 $OUT/p2
