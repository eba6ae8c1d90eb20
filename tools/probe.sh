#!/bin/bash
# what the GPU box really gives a job: cgroup memory / cpu limits, tmpfs sizes
for f in /sys/fs/cgroup/memory.max /sys/fs/cgroup/memory.high /sys/fs/cgroup/memory.current /sys/fs/cgroup/memory.swap.max /sys/fs/cgroup/cpu.max /sys/fs/cgroup/memory/memory.limit_in_bytes /sys/fs/cgroup/memory/memory.usage_in_bytes; do [ -e $f ] && echo "$f: $(cat $f)"; done
cat /proc/self/cgroup | head -3
grep -i "memtotal\|memavailable\|shmem:" /proc/meminfo
df -h /dev/shm /tmp / | cat
ulimit -a | grep -i "mem\|virtual\|locked"
python - <<'PY'
import os
print("sysconf ram GB", os.sysconf("SC_PHYS_PAGES")*os.sysconf("SC_PAGE_SIZE")/1e9, "cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
PY
