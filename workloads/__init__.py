"""Synthetic workloads of BASELINE.json's configurations (detector-line generators, dense maps, seeded weights, the cfg4
homography recipe).  Test / benchmark infrastructure: nothing in the product package `linetr_amd` imports this."""
