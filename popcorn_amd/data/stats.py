"""Per-band normalisation constants (reference data/config/dataset_stats.json: 'sen2springNIR' and 'sen1';
SURVEY.md section 2a).  Model-input channel order is [R, G, B, NIR, VV, VH] (utils/utils.py:171)."""
S2_MEAN = (1460.4567, 1468.2986, 1383.4556, 2226.6821)
S2_STD = (1130.7949, 1129.0261, 1053.3217, 1724.3213)
S1_MEAN = (-11.426, -17.753)
S1_STD = (5.5983, 5.0076)
MEAN6 = S2_MEAN + S1_MEAN
STD6 = S2_STD + S1_STD

# On-"disk" synthetic tile: 13 Sentinel-2 L1C bands [B1,B2,B3,B4,B5,B6,B7,B8,B8A,B9,B10,B11,B12]
# (utils/01_download_gee_country.py:244) followed by Sentinel-1 [VV,VH].  The loader keeps [B4,B3,B2,B8] = R,G,B,NIR
# (data/PopulationDataset.py:566-568) and [VV,VH].
RAW_BANDS = 15
BAND6 = (3, 2, 1, 7, 13, 14)
