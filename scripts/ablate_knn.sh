#!/bin/bash
# experiment helper: rebuild with -D switches, run a dev script, rebuild the real library
flags="$1"; shift
DGCNN_EXTRA_HIPCC_FLAGS="$flags" python "ahn3-dgcnn.pytorch_amd/build.py" --force > /dev/null 2>&1
"$@"
python "ahn3-dgcnn.pytorch_amd/build.py" --force > /dev/null 2>&1
