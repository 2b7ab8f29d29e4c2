for cfg in "dgq 2" "dgp 2" "dgq 1" "dgp 1"; do set -- $cfg
 for alg in auto direct; do
  PDH_FORCE_ALG=$alg timeout -k 10 200 python tools/ab_bench.py --lib-a polydeal_amd/lib/libpolydeal_hip.so --lib-b polydeal_amd/lib/libpolydeal_hip.so --fe $1 --degree $2 --alg $alg --rounds 3 2>&1 | tail -1 | sed "s/^/$1 p=$2 $alg: /"
 done
done
