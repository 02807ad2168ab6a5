"""ImageDataGeneratorCustom (reference: lib/data_generator.py, a thin subclass of keras-preprocessing 1.1.2's
ImageDataGenerator).  Only what the reference's augmentation uses is restated (lib/trainer.py:14-56,
lib/network.py:109-125,149-161): rotation / shift / shear / zoom parameters drawn from NumPy's global
RandomState in keras-preprocessing's order, horizontal / vertical flips, one affine warp per sample with
any of keras-preprocessing's four fill modes ('nearest' -- the reference default --, 'constant' with `cval`, 'reflect', 'wrap');
the warp itself runs on the GPU (pseg_affine_warp_fill: cubic B-spline for the image, nearest for binary and mask), and so does
the brightness shift of `brightness_range` (pseg_brightness_shift: one factor per sample, drawn after the flips; the reference
hands the range to the IMAGE generator only, lib/trainer.py:40-50).  Channel shifts and featurewise statistics raise (no field
of AugmentationSettings reaches them).
keras-preprocessing and the scipy release it ran on are absent offline: parity unpinned (tests compare the
warp with the installed scipy, the parameter stream with its published algorithm)."""
import numpy as np


class ImageDataGeneratorCustom:
    def __init__(self, rotation_range=0, width_shift_range=0., height_shift_range=0., brightness_range=None,
                 shear_range=0., zoom_range=0., channel_shift_range=0., fill_mode='nearest', cval=0.,
                 horizontal_flip=False, vertical_flip=False, rescale=None, preprocessing_function=None,
                 data_format='channels_last', validation_split=0.0, dtype='float32', interpolation_order=1, **unused):
        if data_format != 'channels_last':
            raise Exception("only data_format='channels_last' is built")
        if fill_mode not in ('nearest', 'constant', 'reflect', 'wrap'):
            raise Exception("Unknown fill_mode %r (keras-preprocessing takes 'constant', 'nearest', 'reflect', 'wrap')" % (fill_mode,))
        if channel_shift_range:
            raise Exception("channel shifts are not built (no field of AugmentationSettings sets them)")
        if brightness_range is not None:
            # keras_preprocessing/image/image_data_generator.py: the same check, the same message
            if not isinstance(brightness_range, (tuple, list)) or len(brightness_range) != 2:
                raise ValueError('`brightness_range should be tuple or list of two floats. Received: %s' % (brightness_range,))
        self.brightness_range = brightness_range
        if interpolation_order not in (0, 3):
            raise Exception("interpolation orders 0 (binary, mask) and 3 (image) are built")
        self.rotation_range = rotation_range
        self.width_shift_range = width_shift_range
        self.height_shift_range = height_shift_range
        self.shear_range = shear_range
        if np.isscalar(zoom_range):
            self.zoom_range = [1 - zoom_range, 1 + zoom_range]
        else:
            self.zoom_range = [zoom_range[0], zoom_range[1]]
        self.horizontal_flip = horizontal_flip
        self.vertical_flip = vertical_flip
        self.rescale = rescale
        self.interpolation_order = interpolation_order
        self.fill_mode = fill_mode
        self.cval = cval
        self.dtype = dtype

    # keras_preprocessing/image/image_data_generator.py: get_random_transform (order of the RNG draws kept)
    def get_random_transform(self, img_shape, seed=None):
        if seed is not None:
            np.random.seed(seed)
        theta = np.random.uniform(-self.rotation_range, self.rotation_range) if self.rotation_range else 0
        if self.height_shift_range:
            tx = np.random.uniform(-self.height_shift_range, self.height_shift_range)
            if np.max(self.height_shift_range) < 1:
                tx *= img_shape[0]
        else:
            tx = 0
        if self.width_shift_range:
            ty = np.random.uniform(-self.width_shift_range, self.width_shift_range)
            if np.max(self.width_shift_range) < 1:
                ty *= img_shape[1]
        else:
            ty = 0
        shear = np.random.uniform(-self.shear_range, self.shear_range) if self.shear_range else 0
        if self.zoom_range[0] == 1 and self.zoom_range[1] == 1:
            zx, zy = 1, 1
        else:
            zx, zy = np.random.uniform(self.zoom_range[0], self.zoom_range[1], 2)
        flip_horizontal = (np.random.random() < 0.5) * self.horizontal_flip
        flip_vertical = (np.random.random() < 0.5) * self.vertical_flip
        # (channel_shift_range would draw here; it is 0)  brightness: the LAST draw -- generators without the range (binary,
        # mask) see the same affine parameters under the shared seed
        brightness = None
        if self.brightness_range is not None:
            brightness = np.random.uniform(self.brightness_range[0], self.brightness_range[1])
        return {'theta': theta, 'tx': tx, 'ty': ty, 'shear': shear, 'zx': zx, 'zy': zy,
                'flip_horizontal': flip_horizontal, 'flip_vertical': flip_vertical, 'brightness': brightness}

    @staticmethod
    def affine_matrix(params, h, w):
        """keras_preprocessing/image/affine_transformations.py: apply_affine_transform's matrix -> (2x2, offset),
        or None for the identity."""
        theta, tx, ty, shear, zx, zy = (params[k] for k in ('theta', 'tx', 'ty', 'shear', 'zx', 'zy'))
        m = None
        if theta != 0:
            t = np.deg2rad(theta)
            m = np.array([[np.cos(t), -np.sin(t), 0], [np.sin(t), np.cos(t), 0], [0, 0, 1]])
        if tx != 0 or ty != 0:
            sm = np.array([[1, 0, tx], [0, 1, ty], [0, 0, 1]])
            m = sm if m is None else np.dot(m, sm)
        if shear != 0:
            sh = np.deg2rad(shear)
            hm = np.array([[1, -np.sin(sh), 0], [0, np.cos(sh), 0], [0, 0, 1]])
            m = hm if m is None else np.dot(m, hm)
        if zx != 1 or zy != 1:
            zm = np.array([[zx, 0, 0], [0, zy, 0], [0, 0, 1]])
            m = zm if m is None else np.dot(m, zm)
        if m is None:
            return None
        o_x, o_y = float(h) / 2 + 0.5, float(w) / 2 + 0.5       # transform_matrix_offset_center
        off = np.array([[1, 0, o_x], [0, 1, o_y], [0, 0, 1]])
        rst = np.array([[1, 0, -o_x], [0, 1, -o_y], [0, 0, 1]])
        m = np.dot(np.dot(off, m), rst)
        return m[:2, :2], m[:2, 2]

    def apply_transform(self, x, params):
        """x: (H,W,C) float array -> transformed (H,W,C) float32."""
        from pseg_amd import engine as _eng
        x = np.asarray(x, dtype=np.float32)
        mo = self.affine_matrix(params, x.shape[0], x.shape[1])
        if mo is not None:
            x = np.stack([_eng.affine_warp(x[..., c], mo[0], mo[1], self.interpolation_order, fill_mode=self.fill_mode, cval=self.cval)
                          for c in range(x.shape[2])], axis=-1)
        if params.get('flip_horizontal', False):
            x = x[:, ::-1]
        if params.get('flip_vertical', False):
            x = x[::-1]
        x = np.ascontiguousarray(x)
        if params.get('brightness') is not None:      # after the warp and the flips, as ImageDataGenerator.apply_transform orders them
            x = _eng.brightness_shift(x, params['brightness'])
        return x

    def random_transform(self, x, seed=None):
        return self.apply_transform(x, self.get_random_transform(x.shape, seed))

    def flow(self, x, seed=None, batch_size=1, **unused):
        """NumpyArrayIterator of keras-preprocessing for a batch array (N,H,W,C), batch_size 1 as the reference
        uses it: every batch re-seeds NumPy with seed + batches_seen, permutes the sample order (one sample: no
        draw), then draws the transform parameters."""
        if batch_size != 1:
            raise Exception("flow() is built for batch_size=1 (lib/network.py:151-153)")
        x = np.asarray(x)
        gen = self

        def it():
            seen = 0
            while True:
                if seed is not None:
                    np.random.seed(seed + seen)
                order = np.random.permutation(len(x))
                for j in order:
                    out = gen.apply_transform(x[j], gen.get_random_transform(x[j].shape))
                    if gen.rescale:
                        out = out * gen.rescale
                    seen += 1
                    yield out[None].astype(gen.dtype)
        return it()
