cd /root/repo
for f in "-DKNN_PLAIN_VALU"; do echo "== $f"; DGCNN_EXTRA_HIPCC_FLAGS="$f" python "ahn3-dgcnn.pytorch_amd/build.py" > /dev/null 2>&1; bash scripts/dev_knn_prof.sh "" | head -4; done
python "ahn3-dgcnn.pytorch_amd/build.py" > /dev/null 2>&1
echo "== baseline"; bash scripts/dev_knn_prof.sh "" | head -8
