"""Summarise rocprofv3 output directories written by scripts/profile_gpu.sh:
per-kernel average duration (kernel-trace stats) and PMC counters averaged per
dispatch, for the promonet kernels only."""
import csv
import json
import re
import sys
from collections import defaultdict
from pathlib import Path


def short(name):
    m = re.match(r'void (\w+)<(.*)>\(', name)
    if not m:
        return name.split('(')[0][:60]
    args = m.group(2).replace('Elem', '').replace(' ', '')
    return f'{m.group(1)}<{args}>'


def find(root, suffix):
    return sorted(Path(root).rglob(f'*{suffix}'))


def main(root):
    root = Path(root)
    summary = {}
    for file in find(root / 'stats', 'kernel_stats.csv'):
        rows = list(csv.DictReader(open(file)))
        table = []
        for row in rows:
            name = short(row['Name'])
            table.append({
                'kernel': name, 'calls': int(row['Calls']),
                'avg_us': float(row['AverageNs']) / 1e3,
                'total_ms': float(row['TotalDurationNs']) / 1e6,
                'pct': float(row['Percentage'])})
        summary['kernel_stats'] = table
        print(f'== kernel-trace stats ({file.name}) ==')
        for r in table[:30]:
            print(f"{r['pct']:6.2f}%  calls {r['calls']:4d}  avg "
                  f"{r['avg_us']:10.1f} us  {r['kernel']}")
    for tag in ('pmc_sq', 'pmc_lds', 'pmc_fetch', 'pmc_write'):
        files = find(root / tag, 'counter_collection.csv')
        if not files:
            continue
        sums = defaultdict(lambda: defaultdict(float))
        counts = defaultdict(lambda: defaultdict(int))
        for file in files:
            for row in csv.DictReader(open(file)):
                name = short(row['Kernel_Name'])
                if not ('conv_' in name or 'pm_' in name):
                    continue
                sums[name][row['Counter_Name']] += float(row['Counter_Value'])
                counts[name][row['Counter_Name']] += 1
        table = {
            k: {c: sums[k][c] / counts[k][c] for c in sums[k]} for k in sums}
        summary[tag] = table
        print(f'== {tag}: per-dispatch averages ==')
        for k in sorted(table):
            print(k)
            print('    ' + '  '.join(
                f'{c}={v:.4g}' for c, v in sorted(table[k].items())))
    (root / 'summary.json').write_text(json.dumps(summary, indent=1))


if __name__ == '__main__':
    main(sys.argv[1])
