cd /tmp && export TMPDIR=/tmp
for lib in desc_amd/libdesc_amd.so tools/probes/libdesc_amd_lexp1.so tools/probes/libdesc_amd_lexp2.so; do
  rm -rf /tmp/fprof
  DESC_AMD_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fprof -- python3 $GRAFT_REPO_ROOT/bench.py --workload C4 --steps 3 --warmup 1 --no-cpu-baseline --no-convergence > /tmp/f.log 2>&1
  echo "$lib: $(python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/fprof | grep -i "k_layout_node_dev" | tr '\n' ' ')"
done
