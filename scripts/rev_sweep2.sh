set -e
for U in 4 8 16; do echo "U=$U"; LPAMD_REV_ROWDOT_U=$U timeout -k 10 300 python scripts/revised_bench.py --m 32768 --n 65536 --pivots 200 --warmup 3000; done
echo TILE16; LPAMD_REV_TILE=16 timeout -k 10 300 python scripts/revised_bench.py --m 32768 --n 65536 --pivots 200 --warmup 3000
echo NOLIST; LPAMD_REV_UPDATE_UNROLL=4 timeout -k 10 300 python scripts/revised_bench.py --m 32768 --n 65536 --pivots 200 --warmup 3000
