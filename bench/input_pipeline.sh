#!/bin/bash
# Trainer throughput fed by the host input pipeline against device-resident batches, ResNet-50 224^2 on one GPU:
#   bench/input_pipeline.sh [batch] [workers]
# Host path: synthetic uint8 HWC images -> BatchLoader (worker threads, pinned ring) -> H2D on the copy stream ->
# uint8 -> bf16 NHWC kernel; device path: two batches resident in HBM (--device-data).
B=${1:-2048}; W=${2:-6}
common="--model resnet50 --dataset synthetic --image-size 224 --num-classes 1000 --batch-size $B --optimizer sgd --learning-rate 0.1 --weight-decay 0.0001 --no-progress --steps 24 --warmup-steps 6 --dataset-length $((B*32))"
echo "== host loader ($W worker threads)"; python src/main.py $common --num-workers $W 2>&1 | grep -E "Throughput|Error|error"
echo "== device-resident"; python src/main.py $common --device-data 2>&1 | grep -E "Throughput|Error|error"
