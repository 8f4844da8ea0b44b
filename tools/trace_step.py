"""Print the kernel sequence of one graph-replayed minibatch step from a rocprofv3 kernel trace CSV."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = re.sub(r'acktr::', '', n); n = re.sub(r'void ', '', n)
    return n[:64]
idx = [i for i, r in enumerate(rows) if 'head_step_kernel' in r['Kernel_Name']] or \
    [i for i, r in enumerate(rows) if 'policy_loss_kernel' in r['Kernel_Name']]
i0 = idx[len(idx) // 2]
# step = from the previous clip_update end to this step's clip_update end
j = i0
while 'clip_update' not in rows[j]['Kernel_Name']: j -= 1
k = i0
while 'clip_update' not in rows[k]['Kernel_Name']: k += 1
prev = int(rows[j]['End_Timestamp']); t0 = prev
for r in rows[j + 1:k + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%7.1f us  gap %5.1f  %s' % ((e - s) / 1e3, (s - prev) / 1e3, short(r['Kernel_Name'])))
    prev = e
print('step wall %.1f us, %d kernels' % ((prev - t0) / 1e3, k - j))
