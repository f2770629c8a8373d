#!/bin/bash
# BUILD CONTAINER ONLY.  Stage what tools/gpu_ref_pipelines.sh needs on a GPU box into .refpack/ (git-ignored SCRATCH that
# travels with `gpurun` and is deleted afterwards — reference sources never enter the repository's history):
#   .refpack/open3d_ml_ref.tgz   the Open3D-ML checkout's python packages (ml3d/, scripts/), as ONE archive
#   .refpack/reference_side/     results of `tools/ref_pipelines.py --side reference` (the reference's PyTorch-CPU path on the
#                                oracle ops, YAML sizes), computed here because the GPU box has neither checkout nor time
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
REF="${ML3D_REFERENCE_ROOT:-/root/reference}"
mkdir -p "$ROOT/.refpack/reference_side"
tar czf "$ROOT/.refpack/open3d_ml_ref.tgz" -C "$REF" --exclude='__pycache__' --exclude='*.pyc' ml3d scripts
if [ ! -f "$ROOT/.refpack/reference_side/randlanet_reference.npz" ] || [ "$1" = "--force" ]; then
  (cd /tmp && python "$ROOT/tools/ref_pipelines.py" --side reference --model all --ref "$REF" --out "$ROOT/.refpack/reference_side" \
     2>&1 | tr '\r' '\n' | grep -v 'it/s\]$' > "$ROOT/.refpack/reference_side/reference.log")
fi
ls -la "$ROOT/.refpack" "$ROOT/.refpack/reference_side"
