import csv, sys, collections, re
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != "FETCH_SIZE" or "conv_dma" not in r["Kernel_Name"]: continue
    m = re.search(r'conv_\w+_kernel<([^>]*)>', r["Kernel_Name"])
    a = agg.setdefault(m.group(1), [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in agg.items(): print(sys.argv[1][-30:], k, n, f"fetch x2 per launch = {2*v/n/1024:.1f} MB")
