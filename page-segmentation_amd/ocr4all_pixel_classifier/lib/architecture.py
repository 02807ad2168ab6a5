"""Architecture / Optimizers enums (reference: lib/architecture.py:5-90).

Same member names and string values as the reference.  In scope on the GPU engine: fcn_skip, fcn,
unet, res_unet.  The ImageNet-pretrained backbones (image_res_net, mobile_net, effb0-7) need a
weight download the reference performs at construction (lib/model.py:101,327,374); they are out
of scope and raise on use.
"""
import enum

_ENGINE_ARCHS = ("fcn_skip", "fcn", "unet", "res_unet")


def default_preprocess(x):
    """lib/architecture.py:67-68.  The engine applies the same x/255 on the GPU through a
    256-entry table; this host version exists for API parity."""
    return x / 255.0


class Architecture(enum.Enum):
    FCN_SKIP = 'fcn_skip'
    FCN = 'fcn'
    RES_NET = 'image_res_net'
    RES_UNET = 'res_unet'
    MOBILE_NET = 'mobile_net'
    UNET = 'unet'
    EFFNETB0 = 'effb0'
    EFFNETB1 = 'effb1'
    EFFNETB2 = 'effb2'
    EFFNETB3 = 'effb3'
    EFFNETB4 = 'effb4'
    EFFNETB5 = 'effb5'
    EFFNETB6 = 'effb6'
    EFFNETB7 = 'effb7'

    def __call__(self, *args, **kwargs):
        return self.model()

    @property
    def on_engine(self):
        return self.value in _ENGINE_ARCHS

    def model(self):
        """Returns the engine architecture name (the reference returns a Keras constructor)."""
        if not self.on_engine:
            raise Exception("Architecture %s needs ImageNet weights fetched from the network and is "
                            "not available in the MI355X engine" % self.value)
        return self.value

    def preprocess(self):
        """(preprocess function, rgb flag) -- lib/architecture.py:45-64."""
        if not self.on_engine:
            raise Exception("Architecture %s is not available in the MI355X engine" % self.value)
        return default_preprocess, False


class Optimizers(enum.Enum):
    ADAM = 'adam'
    ADAMAX = 'adamax'
    ADADELTA = 'adadelta'
    ADAGRAD = 'adagrad'
    RMSPROP = 'rmsprop'
    SGD = 'sgd'
    NADAM = 'nadam'

    def __call__(self, *args, **kwargs):
        return self.value
