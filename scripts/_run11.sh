cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1200 python -m pytest tests/test_autograd_gpu.py tests/test_fullsize_gpu.py -q -k "graph_train or full_segment_bf16" -s 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r03/t11_a.log
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29573 tests/ep_ckpt_worker.py > gpurun_out/r03/t11_b.log 2>&1
cat gpurun_out/r03/t11_a.log; grep -v "^\[Gloo\]\|amdgpu.ids" gpurun_out/r03/t11_b.log | tail -40
