#!/bin/bash
cd "$(dirname "$0")/bin"
for v in f6s_stamp f6s_stamp4 f6s_stamp13 f6s_stamp14; do timeout 600 ./$v 256 1008 3129 512 5 | tail -10; done
