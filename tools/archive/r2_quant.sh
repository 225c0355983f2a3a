for rows in 100864 87296 65536 131072; do
for k in outpart fc2part fc1part qkvpart; do
echo "$rows $(python tools/kernel_bench.py $k --iters 30 --rows $rows 2>/dev/null | tail -1)"
done; done
