#!/bin/bash
# GPU BOX.  The reference's OWN scripts/run_pipeline.py (unchanged, from the scratch tarball .refpack/open3d_ml_ref.tgz) driving
# the MI355X-native model classes end to end -- `torch -c <yaml> --split test` once per model family on synthetic dataset
# directories (tools/run_pipeline_e2e.py) -- and the comparison of what run_test leaves on disk with the reference side
# (the checkout's PyTorch-CPU models on the oracle ops, computed in the build container: .refpack/e2e_reference).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
OUT="$ROOT/gpurun_out/e2e"
mkdir -p "$OUT"
rm -rf /tmp/o3dml_ref && mkdir -p /tmp/o3dml_ref
tar xzf .refpack/open3d_ml_ref.tgz -C /tmp/o3dml_ref
{
  echo "== $(date -u +%FT%TZ) scripts/run_pipeline.py x native models on $(python -c 'import torch; print(torch.cuda.get_device_name(0))')"
  (cd /tmp && python "$ROOT/tools/run_pipeline_e2e.py" --side native --family all --ref /tmp/o3dml_ref --work /tmp/ml3d_e2e_full --out "$OUT") || true
  echo "== reference side (build container, PyTorch-CPU + oracle ops):"
  grep "^== \[.*exit" .refpack/e2e_reference/reference.log || true
  python tools/run_pipeline_e2e.py --compare "$OUT" .refpack/e2e_reference
} 2>&1 | tee "$OUT/r05_run_pipeline_e2e.log"
find "$OUT" -name "*.labels" -delete
