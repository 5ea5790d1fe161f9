# N > 1 code paths of bench.py on the one device (gloo backend; RCCL needs a GPU per rank): plumbing only, no scaling number
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02m
mkdir -p $O
cd $R
T="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517"
timeout 600 $T bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --batch 256 --direct-bits 14 > $O/commit_2ranks.json 2> $O/commit_2ranks.err
echo "rc=$?" >> $O/commit_2ranks.err
timeout 600 $T bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --batch 300 --direct-bits default --op verify_batch > $O/verify_2ranks.json 2> $O/verify_2ranks.err
echo "rc=$?" >> $O/verify_2ranks.err
timeout 600 $T bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --direct-bits default --op tiled_msm > $O/tiled_2ranks.json 2> $O/tiled_2ranks.err
echo "rc=$?" >> $O/tiled_2ranks.err
timeout 600 $T bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --batch 256 --direct-bits default --op blob_proof > $O/proof_2ranks.json 2> $O/proof_2ranks.err
echo "rc=$?" >> $O/proof_2ranks.err
tail -2 $O/*.err
B="python bench.py --no-cpu-baseline --no-extra-legs --direct-bits 0"
for rl in 64 128 512; do
LWKZG_REDUCE_LANES=$rl $B > $O/bucket_rl$rl.json 2>> $O/err.txt
done
for sp in 1 2 3 4; do
LWKZG_SPLIT=$sp $B > $O/bucket_split$sp.json 2>> $O/err.txt
done
LWKZG_SPLIT=4 LWKZG_REDUCE_LANES=128 $B > $O/bucket_split4_rl128.json 2>> $O/err.txt
