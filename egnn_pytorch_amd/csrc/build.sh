#!/bin/bash
# Build libegnn_hip.so (gfx950 only) in-tree.  Usage: csrc/build.sh [outdir]
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${1:-$HERE/..}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -Wno-inline-asm -Rpass-analysis=kernel-resource-usage"
# start from an empty object directory: the link step and the spill check below must only ever see this run's outputs
rm -rf "$HERE/obj"
mkdir -p "$HERE/obj"
pids=()
for f in knn_select spatial_order adj_expand linear_hl node_mlp_fused edge_fused edge_pw edge_exact edge_exact_bwd edge_bwd edge_tail node_ops layer_api segment_sum entry_lists global_attn linear_f32 node_prep_f32 fp64 linear_split; do
  EXTRA=""
  # the ranking kernel must reproduce the reference's un-fused ((dx*dx+dy*dy)+dz*dz) bit for bit
  [ "$f" = knn_select ] && EXTRA="-ffp-contract=off"
  ( "$HIPCC" $FLAGS $EXTRA -c "$HERE/$f.hip" -o "$HERE/obj/$f.o" 2> "$HERE/obj/$f.res" ) &
  pids+=($!)
done
# the edge pass a second time for coordinate dimensions other than 3 (compile-time CDM = 8)
( "$HIPCC" $FLAGS -DEGNN_EDGE_GENERIC_C -c "$HERE/edge_fused.hip" -o "$HERE/obj/edge_fused_c.o" 2> "$HERE/obj/edge_fused_c.res" ) &
pids+=($!)
# ... and twice more for the training-mode dropout instantiations beyond the standard layer's (wide heads / other coordinate dimensions)
( "$HIPCC" $FLAGS -DEGNN_EDGE_DROP_TU -c "$HERE/edge_fused.hip" -o "$HERE/obj/edge_fused_d.o" 2> "$HERE/obj/edge_fused_d.res" ) &
pids+=($!)
( "$HIPCC" $FLAGS -DEGNN_EDGE_DROP_TU -DEGNN_EDGE_GENERIC_C -c "$HERE/edge_fused.hip" -o "$HERE/obj/edge_fused_cd.o" 2> "$HERE/obj/edge_fused_cd.res" ) &
pids+=($!)
for p in "${pids[@]}"; do wait "$p" || true; done
for res in "$HERE"/obj/*.res; do
  if [ ! -f "${res%.res}.o" ]; then
    echo "error: $(basename "$res" .res).hip does not compile" >&2
    grep -h -v "remark:" "$res" | head -40 >&2
    exit 1
  fi
done
# compiler errors / warnings (the resource-usage remarks are filtered out)
grep -h -v "remark:\|^ *[0-9]* *|\|^ *|\|\^\|remarks\? generated\|^$" "$HERE"/obj/*.res || true
# Register spills: none inside a kernel's loops (edge_fused.hip::edge_min_blocks).  A source whose kernels use scratch at all is
# compiled to device assembly once more and csrc/check_scratch.py looks at where the scratch accesses sit.
for res in "$HERE"/obj/*.res; do
  if grep -q "ScratchSize \[bytes/lane\]: [1-9]" "$res"; then
    f="$(basename "$res" .res)"
    EXTRA=""; src="$f"
    [ "$f" = edge_fused_c ] && { EXTRA="-DEGNN_EDGE_GENERIC_C"; src=edge_fused; }
    [ "$f" = edge_fused_d ] && { EXTRA="-DEGNN_EDGE_DROP_TU"; src=edge_fused; }
    [ "$f" = edge_fused_cd ] && { EXTRA="-DEGNN_EDGE_DROP_TU -DEGNN_EDGE_GENERIC_C"; src=edge_fused; }
    [ "$f" = knn_select ] && EXTRA="-ffp-contract=off"
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed -Wno-inline-asm $EXTRA -S --cuda-device-only \
        -o "$HERE/obj/$f.s" "$HERE/$src.hip" 2> /dev/null
    if ! python3 "$HERE/check_scratch.py" "$HERE/obj/$f.s"; then
      echo "error: $f spills registers inside a loop" >&2
      exit 1
    fi
    echo "note: $f parks registers in scratch outside its loops ($(grep -h "ScratchSize \[bytes/lane\]: [1-9]" "$res" | sed 's/.*lane\]: \([0-9]*\).*/\1/' | sort -n | tail -1) bytes/lane)"
  fi
done
# the product library, and -- separately -- the test-only reference kernels (include/egnn_hip_ref.h; loaded by tests/_reflib.py)
REF_OBJS=("$HERE/obj/linear_split.o")
PROD_OBJS=()
for o in "$HERE"/obj/*.o; do
  case " ${REF_OBJS[*]} " in *" $o "*) ;; *) PROD_OBJS+=("$o") ;; esac
done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/libegnn_hip.so" "${PROD_OBJS[@]}"
echo "built $OUT/libegnn_hip.so"
REF_OUT="${2:-$HERE/../../tests}"
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$REF_OUT/libegnn_hip_ref.so" "${REF_OBJS[@]}"
echo "built $REF_OUT/libegnn_hip_ref.so (test-only)"
