"""Loss / Monitor enums (reference: lib/metrics.py:115-141).  The loss and metric arithmetic
itself (mean sparse softmax cross-entropy, accuracy, jaccard, dice; lib/metrics.py:8-17,60-85)
belongs to the engine's train step (a later SURVEY 8 row)."""
import enum


class Loss(enum.Enum):
    CATEGORICAL_CROSSENTROPY = 'categorical_crossentropy'
    JACCARD_LOSS = 'jaccard'
    DICE_LOSS = 'dice'
    CATEGORICAL_HINGE = 'categorical_hinge'
    CATEGORCAL_FOCAL = 'categorical_focal'
    DICE_AND_CROSSENTROPY = 'dice_and_crossentropy'

    def __call__(self, *args, **kwargs):
        return self.value


class Monitor(enum.Enum):
    VAL_LOSS = 'val_loss'
    VAL_ACCURACY = 'val_accuracy'
    ACCURACY = 'accuracy'
    LOSS = 'loss'
    DICE_COEF = 'dice_coef'
    JACRAD_COEF = 'jacard_coef'
    FGPA = 'fgpa'
