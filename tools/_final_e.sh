#!/bin/bash
bash tools/_final_b.sh c5b8 c5bdf2
bash tools/_final_d.sh
