#!/usr/bin/env python3
"""FUNSD chargrid training entry point on MI355X -- counterpart of the reference script of the same name
(reference: train_chargrid_funsd_msau.py:16-118 train, :121-163 evaluate, :175-258 main).

Same flow and defaults: pickles written by funsd_preprocessing_word_level.py -> per-document chargrids
(batch 1, variable H x W) -> MSAU(featRoot 8, 4 scales, res_depth 2, softmax) -> Adam(lr 1e-4) with global-norm
clip 1.0 -> accuracy / micro precision / recall over labelled pixels after every epoch -> state_dict saved
every 10 epochs under ckpt/<dataset>_<method>_h<hidden>_o<out>/<epoch>.pth.tar.

Two step implementations, selected with --loop:
  engine     (default) msau_amd.TrainEngine: fused masked-CE + clip + Adam, no per-step host sync
  reference  model(V) -> model.loss -> loss.backward() -> clip_grad_norm_ -> optimizer.step(), line for line
"""
import argparse
import json
import os
import random
import time

import numpy as np
import torch

from msau_amd import MSAUWrapper as MSAU
from msau_amd import TrainEngine
from msau_amd.data import FUNSDCharGridDataLoaderBoxMaskBoxLabel
from msau_amd.training import save_checkpoint


def ckpt_filename(save_dir, args, epoch=-1, isbest=False):
    """utils/io_utils.py:37-80: <dataset>_<method>_h<hidden>_o<out>/[best|<epoch>].pth.tar (epoch 0 -> no number)"""
    d = os.path.join(save_dir, f"{args.bmname or args.dataset}_{args.method}_h{args.hidden_dim}_o{args.output_dim}")
    os.makedirs(d, exist_ok=True)
    if isbest:
        d = os.path.join(d, "best")
    elif epoch > 0:
        d = os.path.join(d, str(epoch))
    return d + ".pth.tar"


def evaluate(dataset, model, args, name="Validation", testing=False, max_num_examples=None, labels_map=None):
    model.eval()
    device = model.flat_parameters.device
    labels, preds = [], []
    with torch.no_grad():
        for batch_idx, data in enumerate(dataset):
            lab = np.squeeze(data["label"].long().numpy())
            _, ypred, _ = model(data["mask"].float().to(device))
            idx = ypred.squeeze(0).argmax(0).cpu().numpy()
            idx = idx[lab != 0]
            lab = lab[lab != 0]
            if testing and labels_map is not None and "other" in labels_map:
                idx[idx == 0] = labels_map["other"]
            labels.append(lab)
            preds.append(idx)
            if max_num_examples is not None and (batch_idx + 1) * args.batch_size > max_num_examples:
                break
    labels, preds = np.hstack(labels), np.hstack(preds)
    acc = float((labels == preds).mean()) if labels.size else 0.0
    # single-label multi-class: micro precision == micro recall == accuracy (sklearn semantics)
    result = {"prec": acc, "recall": acc, "acc": acc}
    print(name, " accuracy:", result["acc"])
    return result


def train(dataset, model, args, val_dataset=None, test_dataset=None, labels_map=None):
    device = model.flat_parameters.device
    if args.loop == "engine":
        engine = TrainEngine(model, lr=args.lr, max_norm=float(args.clip))
        optimizer = None
    else:
        optimizer = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=args.lr)
    best_val = {"epoch": 0, "loss": 0, "acc": 0}
    val_accs = []
    for epoch in range(args.num_epochs):
        t0 = time.time()
        model.train()
        avg_loss = torch.zeros((), device=device)
        print("Epoch: ", epoch)
        for batch_idx, data in enumerate(dataset):
            V = data["mask"].float().to(device)
            label = data["label"].long().to(device)
            if args.loop == "engine":
                loss = engine.step(V, label).reshape(())
            else:
                model.zero_grad()
                _, ypred, ypred_aux = model(V)
                loss = model.loss(ypred, ypred_aux, label)
                loss.backward()
                torch.nn.utils.clip_grad_norm_(model.parameters(), float(args.clip))
                optimizer.step()
            if batch_idx % 10 == 0:
                print("Batch {} optimized. Loss: {}".format(batch_idx, float(loss)))
            avg_loss += loss.detach()
        avg_loss = float(avg_loss) / max(len(dataset), 1)
        print("Avg loss: ", avg_loss, "; epoch time: ", time.time() - t0)
        evaluate(dataset, model, args, name="Train", max_num_examples=100)
        if val_dataset:
            vr = evaluate(val_dataset, model, args, name="Validation")
            val_accs.append(vr["acc"])
            if vr["acc"] > best_val["acc"] - 1e-7:
                best_val = {"acc": vr["acc"], "epoch": epoch, "loss": avg_loss}
        if test_dataset:
            tr = evaluate(test_dataset, model, args, testing=True, name="Test", labels_map=labels_map)
            print("Test result: ", dict(tr, epoch=epoch))
        print("Best val result: ", best_val)
        if epoch % 10 == 0:
            torch.save(model.state_dict(), ckpt_filename(args.ckptdir, args, epoch))
    # final dict checkpoint with the reference's keys (utils/io_utils.py:83-105, called at train_...py:116)
    save_checkpoint(model, engine if args.loop == "engine" else optimizer, args, num_epochs=-1)
    return model, val_accs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-pickle", default="./funsd_preprocess.pkl")
    ap.add_argument("--test-pickle", default="./funsd_preprocess_test.pkl")
    ap.add_argument("--num-epochs", type=int, default=300)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--clip", type=float, default=1.0)          # the reference passes args.clip = True == 1.0
    ap.add_argument("--train-ratio", type=float, default=0.8)
    ap.add_argument("--ckptdir", default="ckpt")
    ap.add_argument("--model-kwargs-path", default=None)
    ap.add_argument("--loop", choices=["engine", "reference"], default="engine")
    ap.add_argument("--dtype", choices=["fp32", "bf16"], default="bf16")
    args = ap.parse_args()
    args.batch_size, args.bmname, args.hidden_dim, args.dataset, args.method = 1, None, 500, "invoice", "GCN"
    random.seed(777)
    data_loader = FUNSDCharGridDataLoaderBoxMaskBoxLabel(args.train_pickle)
    data_loader_test = FUNSDCharGridDataLoaderBoxMaskBoxLabel(args.test_pickle, data_loader.labels)
    args.output_dim = len(data_loader.labels) + 1
    os.makedirs(args.ckptdir, exist_ok=True)
    feature_dim = data_loader[0]["mask"].shape[1]
    if args.model_kwargs_path is None:
        model_kwargs = dict(model="msau", final_act="softmax", featRoot=8, scale_space_num=4, res_depth=2,
                            n_class=args.output_dim, img_channels=feature_dim, use_auxiliary_loss=False)
        with open("model_kwargs.json", "w") as fh:
            json.dump(model_kwargs, fh)
    else:
        with open(args.model_kwargs_path) as fh:
            model_kwargs = json.load(fh)
    model = MSAU(feature_dim, args.output_dim, model_kwargs=dict(model_kwargs, dtype=args.dtype)).cuda()
    indices = list(range(len(data_loader)))
    random.shuffle(indices)
    cut = int(len(indices) * args.train_ratio)
    train_instances = [data_loader[i] for i in indices[:cut]]
    val_instances = [data_loader[i] for i in indices[cut:]]
    test_instances = [data_loader_test[i] for i in range(len(data_loader_test))]
    print("Num training instances: ", len(train_instances), "; Num validation instances: ", len(val_instances),
          "; Num testing instances: ", len(test_instances))
    # batch 1 with a different H x W per document (data_generator_funsd_bert.py:216-222): every shape has its own static
    # plan (buffers + launch list).  Keep them all -- one training and one forward-only plan per distinct shape, bounded
    # by model.max_plan_bytes -- instead of rebuilding a plan on every step of every epoch.
    shapes = {tuple(d["mask"].shape[2:]) for d in train_instances + val_instances + test_instances}
    model.max_cached_plans = max(model.max_cached_plans, 2 * len(shapes))
    print("Distinct document shapes: ", len(shapes))
    train(train_instances, model, args, val_dataset=val_instances, test_dataset=test_instances,
          labels_map=data_loader.labels)
    print("Finished\n\n")


if __name__ == "__main__":
    main()
