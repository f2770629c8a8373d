#!/bin/bash
# GPU BOX.  The reference's own run_inference pipelines + unchanged YAML configs driving the MI355X-native model classes
# (tools/ref_pipelines.py --side native) at the YAML sizes, compared with the reference's PyTorch-CPU path computed in the
# build container (.refpack/reference_side, tools/pack_reference.sh).  Logs -> gpurun_out/pipelines/ (copy to profiles/).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
OUT="$ROOT/gpurun_out/pipelines"
mkdir -p "$OUT"
rm -rf /tmp/o3dml_ref && mkdir -p /tmp/o3dml_ref
tar xzf .refpack/open3d_ml_ref.tgz -C /tmp/o3dml_ref
{
  echo "== $(date -u +%FT%TZ) reference pipelines x native models on $(python -c 'import torch; print(torch.cuda.get_device_name(0))')"
  (cd /tmp && python "$ROOT/tools/ref_pipelines.py" --side native --model all --ref /tmp/o3dml_ref --out "$OUT" 2>&1 | tr '\r' '\n' | grep -v 'it/s\]$' | grep -v '^$')
  echo "== reference side (PyTorch-CPU models + oracle ops, build container):"
  grep -o '\[[a-z]*/reference\].*' .refpack/reference_side/reference.log || true
  python tools/ref_pipelines.py --compare "$OUT" .refpack/reference_side
  echo "== KPFCNN with the GPU-side sampler index (default; sphere order differs from sklearn's unsorted query_radius):"
  (cd /tmp && python "$ROOT/tools/ref_pipelines.py" --side native --model kpconv --sampler-index gpu --ref /tmp/o3dml_ref --out "$OUT/gpu_index" 2>&1 | tr '\r' '\n' | grep '^\[')
  cp .refpack/reference_side/kpconv_reference.npz "$OUT/gpu_index/" 2>/dev/null || true
  python tools/ref_pipelines.py --compare "$OUT/gpu_index" "$OUT/gpu_index" | grep kpconv || true
  echo "   (expected: without the sklearn index the spheres are cut from differently ordered lists -- same SETS, different random subsets; statistically equivalent, not identical)"
} 2>&1 | tee "$OUT/r03_pipeline_run.log"
rm -f "$OUT"/*.npz "$OUT"/gpu_index/*.npz
