for m in 0 1 2 4 8 16 31; do
  if [ $m = 0 ]; then L=""; else L="SS_LIB=stereospike_amd/lib/libss_neuron_dg$m.so"; fi
  echo "== ablate $m"; env $L ONLY=deconv1,deconv2 ROUNDS=2 REPS=3 python tools/bench_upconv_bwd.py 2>&1 | grep "dgrad fused"
done
