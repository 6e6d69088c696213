"""Import path of the reference's TSTR classifier (evaluate/nn_classifier.py): the RNN classifier on the MI355X kernels.
The FCN baseline of that file is outside the path this repository accelerates."""
from timediff_amd.classifier import RNN, nnTimeSeriesClassifier, normalize_data, standardize_data  # noqa: F401
