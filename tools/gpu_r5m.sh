#!/bin/bash
export G2OHIP_LIB=$PWD/variants/chainabl/libg2ohip.so
TAG=chainabl bash tools/gpu_ktrace.sh --opt tree_backward=2 2>&1 | grep -E "chain_backward|front_backward|tree_backward"
