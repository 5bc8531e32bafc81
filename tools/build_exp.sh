#!/bin/bash
# Experiment build of the library in a scratch copy of csrc/ with ONE source edit applied (no -D flags in the product sources):
#   tools/build_exp.sh <tag> <file> <sed expression>   ->  unidefense_amd/libud_exp_<tag>.so   (A/B with tools/gpu_lib_ab.sh <tag>)
set -e
tag=$1; file=$2; expr=$3
root=$(cd "$(dirname "$0")/.." && pwd)
d=/tmp/ud_exp_$tag
rm -rf $d && mkdir -p $d/unidefense_amd $d/include
cp -r $root/unidefense_amd/csrc $d/unidefense_amd/csrc
cp $root/include/*.h $d/include/
rm -f $d/unidefense_amd/csrc/*.o
sed -i "$expr" $d/unidefense_amd/csrc/$file
diff <(cat $root/unidefense_amd/csrc/$file) $d/unidefense_amd/csrc/$file | head -5 || true
make -C $d/unidefense_amd/csrc -j8 > $d/build.log 2>&1 || { tail -20 $d/build.log; exit 1; }
cp $d/unidefense_amd/libunidefense_hip.so $root/unidefense_amd/libud_exp_$tag.so
echo "built unidefense_amd/libud_exp_$tag.so"
