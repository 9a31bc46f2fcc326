# usage (GPU box): bash tools/det_contend.sh <tag> [ENV=VAL ...] : two determinism_check processes at once (GPU contention), 4 Trainers each
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=$1; shift
ARGS=${DET_ARGS:-2 256 32}   # steps, image size, batch
(env "$@" DET_TRAINERS=4 python tools/determinism_check.py $ARGS > gpurun_out/det_${TAG}_b.txt 2>&1 &)
env "$@" DET_TRAINERS=4 python tools/determinism_check.py $ARGS > gpurun_out/det_${TAG}_a.txt 2>&1
sleep 15
echo "== $TAG $@" >> gpurun_out/det_contend.txt
grep -h "all parameter checksums" gpurun_out/det_${TAG}_a.txt gpurun_out/det_${TAG}_b.txt >> gpurun_out/det_contend.txt
