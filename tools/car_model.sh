# One box, one call: the store-only replay, the replay with the kernel's other costs put back (tools/store_duty.hip), the
# kernel itself (bench.py --config C5, M = 1e6) and its workgroup timeline (tools/car_timeline.py, -DRATO_CDIAG=6 build).
O=gpurun_out/r04_k; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -w tools/store_duty.hip -o /tmp/sd
timeout 300 /tmp/sd > $O/store_duty_C5.txt 2>&1
timeout 300 /tmp/sd 1000000 > $O/store_duty_1e6.txt 2>&1
for r in 1 2 3; do python bench.py --config C5 --no-cpu-baseline --no-scp --no-configs --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('C5 shard: kernel %.4f ms  frac %.3f  step %.4f ms' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['ms_per_step']))"; done > $O/kernel_same_box.txt
python bench.py --workload driving --M 1000000 --no-cpu-baseline --no-scp --no-configs --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('M = 1e6: kernel %.4f ms  frac %.3f  step %.4f ms' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['ms_per_step']))" >> $O/kernel_same_box.txt
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=fast -DRATO_CDIAG=6 -I include -I riskaversetrajopt_amd/csrc -o /tmp/cd6.so riskaversetrajopt_amd/csrc/*.hip -ldl
RATO_SAA_LIB=/tmp/cd6.so timeout 300 python tools/car_timeline.py > $O/car_timeline_C5.txt 2>&1
RATO_SAA_LIB=/tmp/cd6.so timeout 300 python tools/car_timeline.py 1000000 > $O/car_timeline_1e6.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=fast -DRATO_CDIAG=4 -I include -I riskaversetrajopt_amd/csrc -o /tmp/cd4.so riskaversetrajopt_amd/csrc/*.hip -ldl
RATO_SAA_LIB=/tmp/cd4.so timeout 300 python tools/car_phases.py > $O/car_phases_C5.txt 2>&1
cat $O/kernel_same_box.txt; grep -A8 "== H" $O/store_duty_C5.txt; grep -A8 "== H" $O/store_duty_1e6.txt; head -4 $O/car_timeline_C5.txt; cat $O/car_phases_C5.txt
