"""Sums rocprofv3 --pmc counter_collection CSVs per kernel and counter.

usage: python tools/pmc_summarise.py OUT.json DIR [DIR ...]
Each DIR is the -d directory of one `rocprofv3 --kernel-trace --pmc ... --output-format csv` pass.
"""
import csv, glob, json, os, sys

def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            with open(f, newline='') as fh:
                seen = {}
                for row in csv.DictReader(fh):
                    k = row['Kernel_Name'].split('(')[0]
                    c = row['Counter_Name']
                    e = acc.setdefault(k, {}).setdefault(c, {'total': 0.0, 'launches': 0})
                    e['total'] += float(row['Counter_Value'])
                    key = (k, c, row['Dispatch_Id'])
                    if key not in seen:
                        seen[key] = 1
                        e['launches'] += 1
                    for extra in ('VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count', 'Scratch_Size', 'LDS_Block_Size', 'Workgroup_Size'):
                        if extra in row:
                            acc[k].setdefault('_launch', {})[extra] = row[extra]
    json.dump(acc, open(out, 'w'), indent=1)
    for k, v in acc.items():
        if not k.startswith('void k_') and not k.startswith('k_'):
            continue
        print(k, v.get('_launch', {}))
        for c, e in v.items():
            if c != '_launch':
                print('   %-28s total %.6g  launches %d  per launch %.6g' % (c, e['total'], e['launches'], e['total'] / max(1, e['launches'])))

if __name__ == '__main__':
    main()
