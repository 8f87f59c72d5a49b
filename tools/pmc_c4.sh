cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 --kernel-trace --output-format csv -d /tmp/pmcc4 -o pmc -- python $R/tools/prof_c4.py 1 > /tmp/pmcc4.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmcc4/**/pmc_counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:40]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVES': n[k] += 1
for k, v in acc.items():
    if 'rollout' in k:
        w = v['SQ_WAVES']
        print(k, 'launches', n[k], 'waves/launch', w / n[k], 'VALU/wave', v['SQ_INSTS_VALU'] / w, 'LDS/wave', v['SQ_INSTS_LDS'] / w, 'SALU/wave', v['SQ_INSTS_SALU'] / w, 'f64 arith/wave', (v['SQ_INSTS_VALU_FMA_F64'] + v['SQ_INSTS_VALU_MUL_F64'] + v['SQ_INSTS_VALU_ADD_F64']) / w)
PY
