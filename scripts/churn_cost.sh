# Cost of frames with landmark churn (features leaving / entering every few frames), measured by the C++ main_sim mirror
# (eqvio_sim: no Python in the loop). Prints the mean loopTimer sections of timing.csv. usage: bash scripts/churn_cost.sh [maxFeatures]
M=${1:-200}
OUT=${TMPDIR:-/tmp}/eqvio_churn_$M
mkdir -p $OUT
./eqvio_amd/lib/eqvio_sim --duration 20 --trajectory sine --numPoints 12000 --numWalls 6 --wallDistance 3 --maxFeatures $M --seed 5 --imuFreq 500 --imageFreq 30 \
   --outputNoise --coordinateChoice InvDepth --fastRiccati 1 --measurementNoise 1.0 --initialPointVariance 4 --outlierThresholdAbs 1e8 --outlierThresholdProb 1e8 \
   --initialBiasOmegaVariance 0.01 --initialBiasAccelVariance 0.01 --initialAttitudeVariance 0.01 --initialPositionVariance 0.01 --initialVelocityVariance 0.01 \
   --output $OUT --quiet > $OUT/log.txt 2>&1
tail -3 $OUT/log.txt
python - <<PY
import csv
rows=list(csv.reader(open("$OUT/timing.csv")))
hdr=[h.strip() for h in rows[0]]
import numpy as np
a=np.array([[float(x) if x.strip() else 0.0 for x in r[:len(hdr)]] for r in rows[1:] if len(r)>=len(hdr)])
print("frames", len(a))
for i,h in enumerate(hdr):
    if h: print(f"  {h:24s} mean {1e6*a[:,i].mean():9.1f} us   median {1e6*np.median(a[:,i]):9.1f} us")
PY
