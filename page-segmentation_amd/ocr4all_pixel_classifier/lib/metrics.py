"""Loss / Monitor enums (reference: lib/metrics.py:115-141).  The loss and metric arithmetic itself (the six
losses of lib/metrics.py:8-112, accuracy, jaccard, dice) runs in the engine's train step (pseg_train_set_loss,
pseg_train_forward_backward); a Loss member called like the reference's (`loss_func()`) yields the name the
engine is configured with."""
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
