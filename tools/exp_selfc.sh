cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/bench.py --no-cpu --config 5 --steps 100 --warmup 50 2>/dev/null | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('c5 on ', b['value']/1e6, b['roofline']['kernel_ms_mean'], b['state_at_end'], b['config']['regime'])"
python $R/bench.py --no-cpu --config 5 --steps 100 --warmup 50 --no-self-collision 2>/dev/null | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('c5 off', b['value']/1e6, b['roofline']['kernel_ms_mean'], b['state_at_end'], b['config']['regime'])"
python $R/bench.py --no-cpu --no-self-collision 2>/dev/null | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('c2 off', b['value']/1e6, b['roofline']['kernel_ms_mean'])"
python $R/bench.py --no-cpu 2>/dev/null | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('c2 on ', b['value']/1e6, b['roofline']['kernel_ms_mean'])"
cd $R && timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py -m gpu -q -x 2>&1 | tail -2
