for spec in "hpcg 16" "hpcg 32" "hpcg 64" "anderson 32" "anderson 64" "anderson 128" "fem 10" "fem 20" "fem 40"; do
  timeout -k 10 120 python tools/trsv_ab.py $spec "tiled=0" "tiled=-1,fresh=1" || exit 1
done
