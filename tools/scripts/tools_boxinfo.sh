#!/bin/bash
python bench.py --steps 10 --warmup 3 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench', 'kernel_ms %.3f' % r['kernel_avg_ms'], 'copy %.0f' % r['box_copy_GBps'])"
rocm-smi --showmemorypartition --showcomputepartition 2>/dev/null | grep -i "partition" | head -4
rocm-smi --showclocks 2>/dev/null | grep -i "mclk\|fclk\|socclk" | head -4
rocm-smi --showpower --showtemp 2>/dev/null | grep -i "power\|junction\|memory" | head -6
rocminfo 2>/dev/null | grep -i "Compute Unit\|Max Clock Freq\|Marketing" | tail -4
cat /sys/class/kfd/kfd/topology/nodes/*/mem_banks/*/properties 2>/dev/null | grep -i "width\|mem_clk" | sort | uniq -c | head -4
hostname
