#!/bin/bash
# timing experiment: rebuild with a -D ablation switch, time, then rebuild the real library
for flags in "$@"; do
  DGCNN_EXTRA_HIPCC_FLAGS="$flags" python "ahn3-dgcnn.pytorch_amd/build.py" --force > /dev/null 2>&1
  DGCNN_KNN_NO_PREPASS=1 python scripts/dev_knn.py 2>&1 | grep "knn B=8 C=64 N=7000 k=30" | sed "s/^/[$flags] /"
done
python "ahn3-dgcnn.pytorch_amd/build.py" --force > /dev/null 2>&1
