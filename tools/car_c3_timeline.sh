hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=fast -DRATO_CDIAG=6 -I include -I riskaversetrajopt_amd/csrc -o /tmp/cd6.so riskaversetrajopt_amd/csrc/*.hip -ldl
for sp in 1 2 3; do echo "== split $sp"; NWG=$((157 * sp)) RATO_CAR_SMALL_SPLIT=$sp RATO_SAA_LIB=/tmp/cd6.so timeout 300 python tools/car_timeline.py 10000 40 2>&1 | grep -v amdgpu.ids | head -12; done
