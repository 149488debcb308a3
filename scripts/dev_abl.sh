cd /root/repo
for f in "-DKNN_ABLATE_ROUNDS" "-DKNN_ABLATE_ROUNDS -DKNN_ABLATE_LDSREAD" "-DKNN_ABLATE_ROUNDS -DKNN_ABLATE_MFMA" "-DKNN_ABLATE_ROUNDS -DKNN_ABLATE_SIGNS" "-DKNN_ABLATE_ROUNDS -DKNN_ABLATE_SIGNS -DKNN_ABLATE_LDSREAD" "-DKNN_ABLATE_ROUNDS -DKNN_ABLATE_SIGNS -DKNN_ABLATE_STAGE_LOADS"; do echo "== $f"; DGCNN_EXTRA_HIPCC_FLAGS="$f" python "ahn3-dgcnn.pytorch_amd/build.py" > /dev/null 2>&1; bash scripts/dev_knn_prof.sh "" | grep "filter16"; done
python "ahn3-dgcnn.pytorch_amd/build.py" > /dev/null 2>&1
