import json, sys
for l in open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/sweep.log'):
    if l.startswith('=='): print(l.strip()); continue
    if l.startswith('{'):
        d = json.loads(l); k = d.get('kernels') or {}
        row = [f"{d['value']:.0f} fps", f"{d['ms_per_step']} ms/step", f"2nd {d.get('value_second_pass')}", f"kern {d.get('kernel_ms_per_step')}"]
        for name in ('analyze', 'mark', 'mark_fused', 'finalize', 'svd'):
            if name in k: row.append(f"{name} {k[name]['avg_launch_ms']} ms {k[name].get('achieved_GBps','')}")
        for name in ('embed_only', 'detect_only'):
            if name in d and 'value' in d[name]: row.append(f"{name} {d[name]['value']:.0f} fps {d[name].get('frac_of_peak')}")
        print('   ', ' | '.join(row), d['payload_bit_exact'])
