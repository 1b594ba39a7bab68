set -e
for n in 16 13 9; do
for v in "0 128" "1 64" "1 128"; do
  set -- $v
  BORNVI_GRAM_TABLES=$1 BORNVI_GRAM_ROWS=$2 timeout -k 10 120 python tools/probes/gram_probe.py $n 7 | sed "s/^/rows=$2 /"
done
done
