# is the container's CPU quota throttling the process?  cgroup statistics around one bench run
show() { for f in /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpu.stat /sys/fs/cgroup/cpuset.cpus.effective /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us /sys/fs/cgroup/cpu/cpu.stat; do [ -r $f ] && { echo "-- $f"; cat $f; }; done; }
echo "nproc: $(nproc)  online: $(getconf _NPROCESSORS_ONLN)"; grep Cpus_allowed_list /proc/self/status
show
"$@"
echo "== after"; show
