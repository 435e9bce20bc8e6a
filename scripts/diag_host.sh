cd $GRAFT_REPO_ROOT
pyspeed() { python3 - <<'PY'
import time
t=time.perf_counter(); s=0
for i in range(3000000): s+=i*i
print("python loop 3M: %.3f s" % (time.perf_counter()-t))
PY
}
echo "== before"; nproc; uptime; pyspeed
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; tail -1 gpurun_out/gpu_tests.log
echo "== after pytest"; uptime; ps -eo pid,ppid,pcpu,pmem,etime,cmd --sort=-pcpu | head -12; pyspeed
ls /dev/shm | head; df -h /dev/shm | tail -1; free -g | head -2
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench4.log 2>&1; python3 -c "
import json
for l in open('gpurun_out/bench4.log'):
    if l.startswith('{\"metric\"'):
        d=json.loads(l); print('bench after pytest', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
sleep 30; echo "== 30 s later"; uptime; pyspeed
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench5.log 2>&1; python3 -c "
import json
for l in open('gpurun_out/bench5.log'):
    if l.startswith('{\"metric\"'):
        d=json.loads(l); print('bench again', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
