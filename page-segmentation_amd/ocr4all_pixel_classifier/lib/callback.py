"""Training progress protocol (reference: lib/callback.py:10-18): the three methods the
frontend implements to show progress."""


class TrainProgressCallback:
    def init(self, total_iters, early_stopping_iters):
        pass

    def update_loss(self, batch: int, loss: float, acc: float):
        pass

    def next_best(self, epoch, acc, n_best):
        pass
