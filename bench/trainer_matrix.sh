#!/bin/bash
# Short end-to-end runs of src/main.py on one GPU over the option matrix a user of the reference is likely to touch
# (data-parallel one-rank group with the own RCCL communicator, host-fed and device-resident data, evaluation,
# checkpoints, gradient accumulation, bf16 wire, reduce-scatter form, recorded steps, transformer models).
# Every run has its own time limit; the script stops at the first failure.   bash bench/trainer_matrix.sh [outdir]
out=${1:-gpurun_out/matrix}
mkdir -p "$out"
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1
port=29600
run() {
  name=$1; shift
  port=$((port + 1))
  MASTER_PORT=$port timeout -k 5 150 python src/main.py "$@" > "$out/$name.out" 2> "$out/$name.err"
  rc=$?
  echo "$name rc=$rc $(grep -E 'Throughput|accuracy' "$out/$name.out" | tr '\n' ' ' | cut -c1-220)"
  [ $rc -eq 0 ] || { tail -15 "$out/$name.err"; exit 1; }
}
S="--dataset synthetic --no-progress"
run r18_default        $S --dataset-length 4096
run r18_dp_host        $S --distributed --dataset-length 4096 --eval
run r18_dp_graph       $S --distributed --dataset-length 4096 --hip-graph
run r50_graph_224      $S --model resnet50 --image-size 224 --batch-size 64 --dataset-length 1920 --device-data --optimizer sgd --hip-graph
run r50_dp_host        $S --distributed --model resnet50 --image-size 224 --batch-size 256 --dataset-length 5120 --num-workers 4 --optimizer sgd --learning-rate 0.05
run r50_dp_accum_bf16  $S --distributed --model resnet50 --image-size 224 --batch-size 128 --dataset-length 2560 --num-workers 4 --optimizer sgd --grad-accum 2 --grad-dtype bf16
run r50_dp_rsag_ckpt   $S --distributed --model resnet50 --image-size 224 --batch-size 128 --dataset-length 2560 --num-workers 4 --optimizer sgd --reduce-algorithm rs_ag --checkpoint-path "${TMPDIR:-/tmp}/pdt_matrix_r50.pt" --save-every 5
run r50_dp_resume      $S --distributed --model resnet50 --image-size 224 --batch-size 128 --dataset-length 2560 --num-workers 4 --optimizer sgd --reduce-algorithm rs_ag --checkpoint-path "${TMPDIR:-/tmp}/pdt_matrix_r50.pt" --resume auto --epochs 2
run vit_dp_device      $S --distributed --model vit_b_16 --image-size 224 --batch-size 128 --dataset-length 2560 --device-data --optimizer adamw --learning-rate 0.0003
run vit_dp_host        $S --distributed --model vit_b_16 --image-size 224 --batch-size 128 --dataset-length 2560 --num-workers 4 --optimizer adamw --learning-rate 0.0003
run gpt2_dp            $S --distributed --model gpt2_medium --batch-size 8 --dataset-length 160 --optimizer adamw --learning-rate 0.0003 --watchdog-s 120
run r50_det_finite     $S --distributed --model resnet50 --image-size 224 --batch-size 64 --dataset-length 1280 --num-workers 2 --optimizer sgd --deterministic --check-finite --keep-checkpoints 3 --checkpoint-path "${TMPDIR:-/tmp}/pdt_matrix_det.pt" --save-every 4
PDT_FORCE_AUTOTUNE=1 run r50_autotune $S --distributed --model resnet50 --image-size 224 --batch-size 128 --dataset-length 2560 --device-data --optimizer sgd --bucket-mb auto
echo ALL-OK
