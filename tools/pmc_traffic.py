"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the conv kernels, per kernel.
usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json>
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of wide
(16 B/lane) coalesced reads -> doubled; WRITE_SIZE is exact; both are in KiB."""
import collections
import csv
import glob
import json
import re
import sys

# every kernel inside bench.py's HIP-event brackets
CONV_FAMILY = ('conv_igemm', 'conv_pw_persist', 'conv_pw_x3', 'conv_pw_h2', 'splitk_epilogue', 'wg_input', 'wg_output', 'wg4_input', 'wg4_output')


def short(name):
    """'void conv_igemm_dma_kernel<64, 64, ...>(ConvParams)' -> 'conv_igemm_dma_kernel<64, 64, ...>'"""
    name = re.sub(r'^void\s+', '', name)
    depth = 0
    for i, ch in enumerate(name):
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            return name[:i]
    return name


def collect(d, counter):
    per = collections.defaultdict(lambda: [0.0, 0])
    eps = 0
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            if any(k in r['Kernel_Name'] for k in CONV_FAMILY):
                e = per[short(r['Kernel_Name'])]
                e[0] += float(r['Counter_Value'])
                e[1] += 1
            elif 'mask_rle_kernel' in r['Kernel_Name']:
                eps += 1
    return per, eps


fetch, ef = collect(sys.argv[1], 'FETCH_SIZE')
write, ew = collect(sys.argv[2], 'WRITE_SIZE')
per_kernel = {}
for name in sorted(set(fetch) | set(write)):
    fk, fn = fetch.get(name, [0.0, 0])
    wk, wn = write.get(name, [0.0, 0])
    per_kernel[name] = {'launches_per_episode': fn / max(ef, 1),
                        'FETCH_SIZE_KiB_per_launch_raw': fk / max(fn, 1), 'WRITE_SIZE_KiB_per_launch': wk / max(wn, 1),
                        'hbm_bytes_per_launch': (2.0 * fk / max(fn, 1) + wk / max(wn, 1)) * 1024.0,
                        'hbm_bytes_per_episode': (2.0 * fk / max(ef, 1) + wk / max(ew, 1)) * 1024.0}
out = {'kernels': ' + '.join(CONV_FAMILY), 'episodes_fetch_pass': ef, 'episodes_write_pass': ew,
       'hbm_bytes_per_episode': sum(v['hbm_bytes_per_episode'] for v in per_kernel.values()),
       'per_kernel': per_kernel,
       'note': 'FETCH_SIZE doubled (gfx950 wide-read correction); separate --pmc passes; same command as bench.py'}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != 'per_kernel'}))
for n, v in per_kernel.items():
    print(f'{n:60s} {v["launches_per_episode"]:7.1f} launches/ep  {v["hbm_bytes_per_launch"] / 1e6:9.2f} MB/launch  {v["hbm_bytes_per_episode"] / 1e9:7.3f} GB/ep')
