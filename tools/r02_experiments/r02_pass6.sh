set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02h
mkdir -p $O
cd $R
B="python bench.py --no-cpu-baseline --no-extra-legs --op blob_proof --caller-streams 2"
for n in 256 1024; do
for hs in 0 1; do
for pr in 0 1; do
for fill in 512 2048 8192; do
LWKZG_HEAVY_SERIAL=$hs LWKZG_HASH_PRIO=$pr LWKZG_DIRECT_FILL=$fill $B --batch $n --steps 12 > $O/p${n}_hs${hs}_pr${pr}_f${fill}.json 2>> $O/err.txt
done; done; done; done
