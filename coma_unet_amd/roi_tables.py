"""FreeSurfer ROI ids and names the model paints and the loss weights
(/root/reference/attn_unet_data_parallel.py:561-597)."""
ROI_INDICES = [
    1001, 1006, 1007, 1009, 1015, 1016, 1030, 1034, 1033, 1008, 1025, 1029, 1031, 1022, 17, 18,
    2001, 2006, 2007, 2009, 2015, 2016, 2030, 2034, 2033, 2008, 2025, 2029, 2031, 2022, 49, 50, 51, 52, 53, 54,
]
ROI_NAMES = [
    'ctx-lh-bankssts', 'ctx-lh-entorhinal', 'ctx-lh-fusiform', 'ctx-lh-inferiortemporal',
    'ctx-lh-middletemporal', 'ctx-lh-parahippocampal', 'ctx-lh-superiortemporal',
    'ctx-lh-transversetemporal', 'ctx-lh-temporalpole', 'ctx-lh-inferiorparietal',
    'ctx-lh-precuneus', 'ctx-lh-superiorparietal', 'ctx-lh-supramarginal', 'ctx-lh-postcentral',
    'Left-Hippocampus', 'Left-Amygdala', 'ctx-rh-bankssts', 'ctx-rh-entorhinal',
    'ctx-rh-fusiform', 'ctx-rh-inferiortemporal', 'ctx-rh-middletemporal',
    'ctx-rh-parahippocampal', 'ctx-rh-superiortemporal', 'ctx-rh-transversetemporal',
    'ctx-rh-temporalpole', 'ctx-rh-inferiorparietal', 'ctx-rh-precuneus',
    'ctx-rh-superiorparietal', 'ctx-rh-supramarginal', 'ctx-rh-postcentral', 'Right-Thalamus-Proper',
    'Right-Caudate', 'Right-Putamen', 'Right-Pallidum', 'Right-Hippocampus', 'Right-Amygdala',
]
# NOTE the reference's name LIST (:567-578) and its index->name DICT (:581-597) disagree on order
# (e.g. 1034 vs 1033); forward() uses the DICT (:642), so the dict is authoritative here.
ROI_INDEX_TO_NAME = {
    1001: 'ctx-lh-bankssts', 1006: 'ctx-lh-entorhinal', 1007: 'ctx-lh-fusiform',
    1009: 'ctx-lh-inferiortemporal', 1015: 'ctx-lh-middletemporal',
    1016: 'ctx-lh-parahippocampal', 1030: 'ctx-lh-superiortemporal',
    1034: 'ctx-lh-transversetemporal', 1033: 'ctx-lh-temporalpole',
    1008: 'ctx-lh-inferiorparietal', 1025: 'ctx-lh-precuneus',
    1029: 'ctx-lh-superiorparietal', 1031: 'ctx-lh-supramarginal', 1022: 'ctx-lh-postcentral',
    17: 'Left-Hippocampus', 18: 'Left-Amygdala', 2001: 'ctx-rh-bankssts',
    2006: 'ctx-rh-entorhinal', 2007: 'ctx-rh-fusiform', 2009: 'ctx-rh-inferiortemporal',
    2015: 'ctx-rh-middletemporal', 2016: 'ctx-rh-parahippocampal',
    2030: 'ctx-rh-superiortemporal', 2034: 'ctx-rh-transversetemporal',
    2033: 'ctx-rh-temporalpole', 2008: 'ctx-rh-inferiorparietal',
    2025: 'ctx-rh-precuneus', 2029: 'ctx-rh-superiorparietal', 2031: 'ctx-rh-supramarginal',
    2022: 'ctx-rh-postcentral', 49: 'Right-Thalamus-Proper', 50: 'Right-Caudate',
    51: 'Right-Putamen', 52: 'Right-Pallidum', 53: 'Right-Hippocampus',
    54: 'Right-Amygdala',
}
