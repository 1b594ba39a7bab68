set -e
for v in shipped plain shipped plain; do
  if [ $v = shipped ]; then unset BORNVI_LIB; else export BORNVI_LIB=$PWD/tools/_variants/libbornvi_gram_$v.so; fi
  timeout -k 10 120 python tools/probes/gram_probe.py 16 7
done
