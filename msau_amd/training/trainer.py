"""`Trainer` counterpart (reference: model/training/trainer.py:12-207).

Same constructor and `train(data_provider, output_path, restore_path, batch_steps_per_epoch, epochs, ...)`
contract: `data_provider.next_data('train'|'val') -> (x [B,C,H,W], one-hot tgt, one-hot aux tgt)`,
`.size_val`, `.batchsize_tr`, `.restart_val_runner()`, `.stop_all()`; lr = 1e-3 * 0.95^(epoch // 10); a
checkpoint `<output_path>/model<epoch>` whenever the validation loss improves or every 8th epoch.
The network is the HIP `MSAUWrapper`; tensors go to the model's device instead of a hard-coded `.cuda()`."""
from __future__ import annotations

import os
import time

import torch

from .cost import UNetLoss
from .optimizer import get_optimizer


class Trainer:
    def __init__(self, net, opt_kwargs={}, cost_kwargs={}):
        self.net = net
        self.opt_kwargs = opt_kwargs
        self.use_auxiliary_loss = cost_kwargs.get("use_auxiliary_loss", True)
        self.cost_kwargs = {"aux_logits": None, "aux_tgt": None} if self.use_auxiliary_loss else dict(cost_kwargs)
        self.cost_type = cost_kwargs.get("cost_name", "cross_entropy")
        self.criterion = UNetLoss(self.cost_kwargs)

    def _initialize(self, output_path):
        self.optimizer = get_optimizer(self.net, self.opt_kwargs)
        if output_path is not None:
            os.makedirs(os.path.abspath(output_path), exist_ok=True)

    def adjust_lr(self, epoch):
        lr = 0.001 * (0.95 ** (epoch // 10))
        for group in self.optimizer.param_groups:
            group["lr"] = lr
        return lr

    def load_weights(self, weights_dict, output_path):
        self.net.load_weights(weights_dict)
        self.net.save(os.path.join(output_path, "model") + "02")

    def _batch(self, data_provider, split):
        dev = self.net.flat_parameters.device
        bx, bt, ba = data_provider.next_data(split)
        if bx is None:
            return None, None, None
        return bx.float().to(dev), bt.long().to(dev), ba.long().to(dev)

    def _loss(self, bx, bt, ba):
        _, logits, logits_aux = self.net(bx)
        self.cost_kwargs["aux_logits"] = logits_aux if self.use_auxiliary_loss else None
        self.cost_kwargs["aux_tgt"] = ba
        return self.criterion(logits, bt, self.cost_kwargs)

    def train(self, data_provider, output_path, restore_path=None, batch_steps_per_epoch=1024, epochs=250,
              gpu_device="0", max_spat_dim=5000000):
        print("Epochs: " + str(epochs))
        print("Batch Size Train: " + str(data_provider.batchsize_tr))
        print("Batchsteps per Epoch: " + str(batch_steps_per_epoch))
        save_path = os.path.join(output_path, "model") if output_path is not None else None
        if epochs == 0:
            return save_path
        self._initialize(output_path)
        if restore_path is not None:
            print("Loading Checkpoint.")
            self.net.load_weights(restore_path)
        best = 100000.0
        shown = 0
        for epoch in range(epochs):
            lr = self.adjust_lr(epoch)
            tot = tot_final = 0.0
            accs = []
            t0 = time.time()
            self.net.train()
            for _ in range(batch_steps_per_epoch):
                bx, bt, ba = self._batch(data_provider, "train")
                if bx is None:
                    print("No Training Data available. Skip Training Path.")
                    break
                skipped = 0
                while bx.shape[2] * bx.shape[3] > max_spat_dim:
                    bx, bt, ba = self._batch(data_provider, "train")
                    skipped += 1
                    if skipped > 100:
                        print("Spatial Dimension of Training Data to high. Aborting.")
                        return save_path
                self.optimizer.zero_grad()
                acc, loss, final_loss = self._loss(bx, bt, ba)
                accs.append(acc)
                loss.backward()
                self.optimizer.step()
                if final_loss is not None:
                    tot_final += float(final_loss)
                shown += bx.shape[0]
                tot += float(loss)
            self.output_epoch_stats_train(epoch + 1, sum(accs) / max(len(accs), 1), tot / batch_steps_per_epoch,
                                          tot_final / batch_steps_per_epoch, shown, lr, time.time() - t0)
            tot = tot_final = 0.0
            accs = []
            t0 = time.time()
            self.net.eval()
            val_size = data_provider.size_val
            with torch.no_grad():
                for _ in range(val_size):
                    bx, bt, ba = self._batch(data_provider, "val")
                    if bx is None:
                        print("No Validation Data available. Skip Validation Path.")
                        break
                    acc, loss, final_loss = self._loss(bx, bt, ba)
                    accs.append(acc)
                    if final_loss is not None:
                        tot_final += float(final_loss)
                    tot += float(loss)
            if val_size != 0:
                tot, tot_final = tot / val_size, tot_final / val_size
                self.output_epoch_stats_val(epoch + 1, sum(accs) / max(len(accs), 1), tot, tot_final, time.time() - t0)
                data_provider.restart_val_runner()
            if output_path is not None and (tot < best or (epoch + 1) % 8 == 0):
                best = min(best, tot)
                print("Saving checkpoint")
                self.net.save(save_path + str(epoch + 1))
        data_provider.stop_all()
        print("Optimization Finished!")
        print("Best Val Loss: " + str(best))
        return save_path

    def output_epoch_stats_train(self, epoch, acc, total_loss, total_loss_final, shown_sample, lr, time_used):
        print("TRAIN: Epoch {:}, Acc: {:.6f}, Average loss: {:.6f} final: {:.6f}, training samples shown: {:}, "
              "learning rate: {:.6f}, time used: {:.2f}".format(epoch, acc, total_loss, total_loss_final, shown_sample,
                                                                lr, time_used))

    def output_epoch_stats_val(self, epoch, acc, total_loss, total_loss_final, time_used):
        print("VAL: Epoch {:}, Acc: {:.6f}, Average loss: {:.6f} final: {:.6f}, time used: {:.2f}".format(
            epoch, acc, total_loss, total_loss_final, time_used))
