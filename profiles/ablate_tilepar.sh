# needs an ablation build as the in-tree library: bash profiles/build_variants.sh capi.hip "abl:-DMNE_ABLATION" && cp profiles/_variants/lib_abl.so mneslam_amd/libmneslam_hip.so
# per-kernel times of the fused iteration under timing-ablation flags (results are wrong when flags != 0)
mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
out=gpurun_out/ablate_tilepar.txt; : > $out
for f in ${FLAGS:-0 1 2 4 512 4096 1024 5635}; do
  rm -rf /tmp/pa_$f
  MNE_DBG_FLAGS=$f timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pa_$f -o t -- python bench.py --steps 60 --warmup 10 --cpu-iters 0 > /dev/null 2>&1
  db=$(find /tmp/pa_$f -name '*.db' | head -1)
  echo "FLAGS=$f" >> $out
  python profiles/summarize_rocprof_db.py $db 2>&1 | grep -E "backward_kernel|decode_kernel|tile_adam_kernel|wgrad_fused|composite" | cut -c1-150 >> $out
done
cat $out
