# build libpulpo_hip_<tag>.so with one source recompiled under extra -D flags (diagnostic / ablation builds; never loaded by default):
#   bash scripts/build_variant.sh <tag> <source.hip> -DPULPO_PW_ABL=1 ...    then    PULPO_HIP_LIB=$PWD/pulpo_amd/csrc/libpulpo_hip_<tag>.so python ...
set -e
cd "$(dirname "$0")/.."
TAG=$1; SRC=$2; shift 2
python -m pulpo_amd.build > /dev/null 2>&1
O=pulpo_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fvisibility=hidden -Wno-inline-asm "$@" -c pulpo_amd/csrc/$SRC.hip -o /tmp/${SRC}_$TAG.o
OBJS=$(ls $O/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pulpo_amd/csrc/libpulpo_hip_$TAG.so $OBJS /tmp/${SRC}_$TAG.o
echo pulpo_amd/csrc/libpulpo_hip_$TAG.so
