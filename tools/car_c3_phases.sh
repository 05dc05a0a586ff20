# C3 (driving M = 1e4, S = 40): phases of car_linearize_rows_kernel per workgroup (-DRATO_CDIAG=4: prologue / staging /
# rollout / rows; =5: the same without the Jacobian stores) for the tile split 1 / 2, and the kernel itself.
O=gpurun_out/r05_c3; mkdir -p $O
for d in 4 5; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=fast -DRATO_CDIAG=$d -I include -I riskaversetrajopt_amd/csrc -o /tmp/cd$d.so riskaversetrajopt_amd/csrc/*.hip -ldl
done
for sp in 1 2; do
  for d in 4 5; do
    echo "== split $sp  CDIAG $d"
    NWG=$((157 * sp)) RATO_CAR_SMALL_SPLIT=$sp RATO_SAA_LIB=/tmp/cd$d.so timeout 300 python tools/car_phases.py 10000 40 2>&1 | grep -v amdgpu.ids
  done
done > $O/car_phases_C3.txt
for sp in 1 2 3; do
  RATO_CAR_SMALL_SPLIT=$sp python bench.py --config C3 --no-cpu-baseline --no-scp --no-configs --steps 300 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('C3 split $sp: kernel %.4f ms  frac %.3f  step %.4f ms' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['ms_per_step']))"
done > $O/kernel_C3.txt
cat $O/car_phases_C3.txt $O/kernel_C3.txt
