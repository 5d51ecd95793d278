# same-box A/B of attention kernel variants (dev tool): kernel durations from rocprofv3
cd /tmp && export TMPDIR=/tmp
for v in 0 1 0 1; do
  CODA_ATTN_DKV_DB=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -o run -- python /root/repo/tools/bench_attn.py > /dev/null 2>&1
  python3 - $v <<'PY'
import csv,sys
for r in csv.DictReader(open(f'/tmp/ab_{sys.argv[1]}/run_kernel_stats.csv')):
    if 'mha_' in r['Name'] and 'delta' not in r['Name'] and "dkv" in r["Name"]:
        n=r['Name'].split('::')[-1].split('(')[0]
        print(f"DB={sys.argv[1]} {n:50s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us min {float(r['MinNs'])/1e3:8.1f}")
PY
done
