import csv, glob, collections
for d in ('fe_pmc1', 'fe_pmc2', 'fe_pmc3'):
    for path in glob.glob(f'gpurun_out/{d}/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if 'frontend' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in agg.items():
            print(d, k, f"{sum(v) / len(v):.4g}")
