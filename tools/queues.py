"""Which HIP work lands on which hardware queue: rocprofv3 kernel trace of bench.py -> per Queue_Id the launches of a few
role-marking kernels (caller streams: persistent GEMM / det_post; side stream: class_vector / roi_align_mask; upload
stream: dense_rle + H2D blits; copy stream: D2H blits) and the busy time.  usage: queues.py <trace dir>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
roles = {'persistGEMM': 'conv_pw_persist', 'det_post': 'det_post_kernel', 'class_vector(side)': 'class_vector_kernel',
         'roi_align_mask(side)': 'roi_align_mask_kernel', 'dense_rle(upload)': 'dense_rle_walk_kernel', 'blit': '__amd_rocclr_copyBuffer',
         'mask_rle(main end)': 'mask_rle_kernel', 'wg4_input': 'wg4_input_kernel', 'fill': 'FillFunctor', 'cat': 'CatArrayBatchedCopy'}
per = collections.defaultdict(lambda: collections.Counter())
busy = collections.Counter()
for r in rows:
    q = r['Queue_Id']
    busy[q] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    per[q]['all'] += 1
    for k, pat in roles.items():
        if pat in r['Kernel_Name']:
            per[q][k] += 1
for q in sorted(per, key=lambda q: -busy[q]):
    print(f'queue {q}: {per[q]["all"]} kernels, busy {busy[q] / 1e6:.1f} ms :', {k: v for k, v in per[q].items() if k != 'all'})
