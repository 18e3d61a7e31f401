cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in _base ""; do for args in "--workload cfg5" "--workload cfg5 --niter 100 --tol 1e-300" "--workload cfg4" "--workload cfg4 --niter 100 --tol 1e-300" "--workload cfg3" "--workload cfg3 --niter 100 --tol 1e-300" "--workload cfg2" "--workload cfg2 --niter 100 --tol 1e-300"; do
LDSR_HIP_SO=$PWD/ldsr_amd/libldsr_hip$v.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$args', '%.4f ms' % d['roofline']['kernel_ms'], d['roofline']['kernel'], d['config']['units_per_step'])"
done; done
