cd corona-13_amd/csrc; cp libcorona_mi.so /tmp/cur.so
for v in cur t4 t12 cur; do
  if [ $v != cur ]; then cp libcorona_mi_$v.so libcorona_mi.so; else cp /tmp/cur.so libcorona_mi.so; fi
  for c in cfg3 media_ptdl fog_ptdl; do (cd ../..; echo $v $c $(python3 bench.py --no-cpu-baseline --steps 4 --config $c 2>/dev/null | grep -o '"value": [0-9.]*')); done
done
cp /tmp/cur.so libcorona_mi.so
