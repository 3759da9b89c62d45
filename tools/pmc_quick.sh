#!/bin/bash
# instruction / busy counters of every kernel of a few layer forwards: bash tools/pmc_quick.sh <outdir> <shape> [env...]
OUT="$(pwd)/$1"; SHAPE="$2"; shift 2
mkdir -p "$OUT"
REPO="$(pwd)"
cat > /tmp/pmc_run.py <<PY
import sys, torch
sys.path.insert(0, "$REPO")
torch.set_grad_enabled(False)
sys.argv = ["pw_probe", "shapes=$SHAPE", "reps=2"]
exec(open("$REPO/tools/pw_probe.py").read())
PY
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "VALUBusy" "MfmaUtil" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  env "$@" rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pmc$i" -o pmc --output-format csv -- python /tmp/pmc_run.py > "$OUT/pmc$i.log" 2>&1
  echo "pmc$i rc=$?"
done
cd "$REPO"
python tools/summarize_prof.py "$OUT" "$SHAPE" > "$OUT/summary.txt" 2>&1
grep "edge_fused\|edge_pw\|edge_kernel" "$OUT/summary.txt" | cut -c1-160
