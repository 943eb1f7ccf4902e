#!/bin/bash
# Kernel-time profile of deterministic mode: bench/det_profile.sh
export PDT_DETERMINISTIC=1
exec bash bench/profile_step.sh det_resnet50
