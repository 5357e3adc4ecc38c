import csv, collections, glob, sys
tag = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ''
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(f'gpurun_out/pmc_{tag}_*/pmc_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void (anonymous namespace)::','')[:50] + ' g=' + r['Grid_Size']
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k, c in agg.items():
    if filt not in k: continue
    print(k, 'avg dur us %.1f' % (sum(dur[k])/len(dur[k])))
    for name, v in sorted(c.items()): print('   %-30s %.4g' % (name, sum(v)/len(v)))
