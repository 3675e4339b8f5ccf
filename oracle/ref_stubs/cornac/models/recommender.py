"""stand-in for the base classes the compiled BPR class derives from (only construction is needed here)"""
MEASURE_L2 = "l2 distance aka. Euclidean distance"
MEASURE_DOT = "dot product aka. inner product"
MEASURE_COSINE = "cosine similarity"


class Recommender:
    def __init__(self, name, trainable=True, verbose=False):
        self.name = name
        self.trainable = trainable
        self.verbose = verbose


class ANNMixin:
    pass
