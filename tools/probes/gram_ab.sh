set -e
for n in 16 13 8; do
for t in 0 1; do
  BORNVI_GRAM_TABLES=$t timeout -k 10 120 python tools/probes/gram_probe.py $n 5
done
done
