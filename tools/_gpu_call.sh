bash tools/gpu_round_profiles.sh r04 2>&1 | tail -40
