#!/bin/bash
# GPU BOX.  The reference's OWN scripts/run_pipeline.py (unchanged, from the scratch tarball .refpack/open3d_ml_ref.tgz) TRAINING the
# MI355X-native model classes: `torch -c <yaml> --split train`, one epoch of a few optimisation steps + a validation pass + the
# checkpoint, for randlanet_semantic3d.yml and kpconv_semantic3d.yml on a synthetic Semantic3D directory
# (tools/run_pipeline_e2e.py --split train) -- compared with the reference side (the checkout's PyTorch-CPU models on the oracle
# ops, computed in the build container: .refpack/e2e_train_reference): the loss of every step, the epoch summary, the checkpoints.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
OUT="$ROOT/gpurun_out/e2e_train"
mkdir -p "$OUT"
rm -rf /tmp/o3dml_ref && mkdir -p /tmp/o3dml_ref
tar xzf .refpack/open3d_ml_ref.tgz -C /tmp/o3dml_ref
{
  echo "== $(date -u +%FT%TZ) scripts/run_pipeline.py --split train x native models on $(python -c 'import torch; print(torch.cuda.get_device_name(0))')"
  (cd /tmp && python "$ROOT/tools/run_pipeline_e2e.py" --side native --family all --split train --ref /tmp/o3dml_ref --work /tmp/ml3d_e2e_trfull --out "$OUT") || true
  echo "== reference side (build container, PyTorch-CPU + oracle ops):"
  grep "^== \[.*exit" .refpack/e2e_train_reference.log || true
  python tools/run_pipeline_e2e.py --compare "$OUT" .refpack/e2e_train_reference
} 2>&1 | tee "$OUT/r04_run_pipeline_train_e2e.log"
find "$OUT" -name "*.pth" -delete
