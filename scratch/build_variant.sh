# A development build of the library with extra flags on SOME sources: bash scratch/build_variant.sh NAME "FLAGS" src1.hip [src2.hip ...]
# -> scratch/libs/lib_NAME.so (the other objects come from the product build under csrc/_obj); select it with ISX_LIB=scratch/libs/lib_NAME.so
set -e
NAME=$1; FLAGS=$2; shift 2
R=$(cd $(dirname $0)/.. && pwd); C=$R/iterseg_amd/csrc; mkdir -p $R/scratch/libs /tmp/var_$NAME
python3 -c "from iterseg_amd import _build; _build.build()" 
OBJS=""
for o in $C/_obj/*.o; do
  b=$(basename $o .o); use=$o
  for s in "$@"; do if [ "$s" = "$b.hip" ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt $(python3 -c "from iterseg_amd import _build; print(' '.join(_build.EXTRA.get('$s', [])))") $FLAGS -c $C/$s -o /tmp/var_$NAME/$b.o
    use=/tmp/var_$NAME/$b.o
  fi; done
  OBJS="$OBJS $use"
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/scratch/libs/lib_$NAME.so $OBJS -ldl
echo built scratch/libs/lib_$NAME.so
