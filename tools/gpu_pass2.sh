#!/bin/bash
bash tools/ab.sh "full_sup2 full_nosup" 4
bash tools/ab.sh "full_nosup full_sup2" 3
