class DataCounter:
    """Iteration / epoch / batch bookkeeping (eeyore/datasets/data_counter.py:1-80).  ``num_batches == 1``
    selects the samplers' cached-target fast path (eeyore/samplers/hmc.py:129,150)."""

    def __init__(self, batch_size, sample_size, num_epochs=None, num_burnin_epochs=None, num_batches=None,
                 drop_last=False):
        self.set_data_info(batch_size, sample_size, num_batches=num_batches, drop_last=drop_last)
        self.set_epoch_info(num_epochs, num_burnin_epochs)
        self.reset()

    def set_num_batches(self, drop_last=False):
        self.num_batches = self.sample_size // self.batch_size
        if (self.sample_size % self.batch_size != 0) and not drop_last:
            self.num_batches = self.num_batches + 1

    def set_data_info(self, batch_size, sample_size, num_batches=None, drop_last=False):
        self.batch_size = batch_size
        self.sample_size = sample_size
        if num_batches is None:
            self.set_num_batches(drop_last=drop_last)
        else:
            self.num_batches = num_batches

    def set_data_info_from_dataloader(self, dataloader):
        self.set_data_info(dataloader.batch_size, len(dataloader.dataset), num_batches=len(dataloader))

    def set_num_iters(self, num_epochs):
        self.num_epochs = num_epochs
        self.num_iters = None if num_epochs is None else num_epochs * self.num_batches

    def set_num_burnin_iters(self, num_burnin_epochs):
        self.num_burnin_epochs = num_burnin_epochs
        self.num_burnin_iters = None if num_burnin_epochs is None else num_burnin_epochs * self.num_batches

    def set_epoch_info(self, num_epochs, num_burnin_epochs):
        self.set_num_iters(num_epochs)
        self.set_num_burnin_iters(num_burnin_epochs)

    def set_num_epochs(self, num_iters):
        self.num_iters = num_iters
        if num_iters is None:
            self.num_epochs = None
        else:
            self.num_epochs = -(-num_iters // self.num_batches)

    def set_num_burnin_epochs(self, num_burnin_iters):
        self.num_burnin_iters = num_burnin_iters
        if num_burnin_iters is None:
            self.num_burnin_epochs = None
        else:
            self.num_burnin_epochs = -(-num_burnin_iters // self.num_batches)

    def set_iter_info(self, num_iters, num_burnin_iters):
        # the reference passes `self` twice here (data_counter.py:62-64) and raises; this is the intended behaviour
        self.set_num_epochs(num_iters)
        self.set_num_burnin_epochs(num_burnin_iters)

    @classmethod
    def from_dataloader(selfclass, dataloader, num_epochs=None, num_burnin_epochs=None):
        return selfclass(dataloader.batch_size, len(dataloader.dataset), num_epochs=num_epochs,
                         num_burnin_epochs=num_burnin_epochs, num_batches=len(dataloader))

    def reset(self):
        self.idx = 0

    def increment_idx(self, incr=1):
        self.idx = self.idx + incr
