import csv,glob,re,sys
from collections import defaultdict
pat=sys.argv[1] if len(sys.argv)>1 else 'gemm_h3'
tot=defaultdict(lambda: defaultdict(float)); cnt=defaultdict(lambda: defaultdict(set))
for f in glob.glob('/root/repo/gpurun_out/pmc_lab/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=re.sub(r'\(.*$','',r['Kernel_Name'])
        if pat not in k: continue
        tot[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']].add(r['Dispatch_Id'])
for k in tot:
    print(k)
    for c,v in sorted(tot[k].items()):
        print('   %-45s %16.0f per launch'%(c, v/len(cnt[k][c])))
