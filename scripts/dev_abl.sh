cd /root/repo
for f in "-DRF_ABLATE_FINAL" "-DRF_ABLATE_INSERT -DRF_ABLATE_FINAL"; do echo "== $f"; DGCNN_EXTRA_HIPCC_FLAGS="$f" python "ahn3-dgcnn.pytorch_amd/build.py" > /dev/null 2>&1; bash scripts/dev_knn_prof.sh "" | grep "refine"; done
python "ahn3-dgcnn.pytorch_amd/build.py" > /dev/null 2>&1
