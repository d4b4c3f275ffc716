#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/gpu_check_train.sh
bash tools/gpu_pmc_train.sh
